// tables.h — host-built constant tables for the Mode-S kernels.
//
// The tables are small, built once per context on the host and uploaded; building them on
// the host with plain IEEE float/integer arithmetic is what makes the UC8 magnitudes and the
// syndrome lookups bit-identical to the reference (convert.c:35-62, crc.c:42-64,180-350).
#pragma once
#include <cstdint>
#include <vector>

namespace mgpu {

// CRC-24 of Mode S, generator 0xFFF409 (crc.c:31)
struct CrcTables {
    uint32_t byte_table[256];   // remainder of each single-byte message (crc.c:42-57)
    uint32_t bit_syndrome[112]; // syndrome of a single wrong bit k of a 112-bit frame (crc.c:59-64);
                                // bit k of a 56-bit frame has syndrome bit_syndrome[k + 56]
    CrcTables();
    uint32_t checksum(const uint8_t *msg, int bits) const;  // modesChecksum, crc.c:67-82
};
const CrcTables &crc_tables();

// One correctable error pattern: syndrome -> up to two bit positions (struct errorinfo, crc.h:32-38)
struct SyndromeEntry {
    uint32_t syndrome;
    int8_t nerr;      // 1 or 2
    int8_t bit0, bit1;
};

// Sorted, collision-free syndrome table for `bits`-long frames (prepareErrorTable, crc.c:180-350):
// nfix 1 -> (max_correct 1, max_detect 1), nfix 2 -> (2, 4), as modesChecksumInit (crc.c:353-378).
std::vector<SyndromeEntry> build_syndrome_table(int bits, int nfix);

// Device-side packing of one entry: syndrome<<16 | bit0<<8 | bit1 (bit1 = 0xFF when nerr == 1),
// sorted ascending, so one 8-byte compare finds an entry.
std::vector<uint64_t> pack_syndrome_table(const std::vector<SyndromeEntry> &t);

// Parity masks for the wave-parallel CRC: syndrome bit j of a frame held as
//   hi = frame bits 0..63   (bit 0 at bit 63 of the word)
//   lo = frame bits 64..111 (bit 64 at bit 47 of the word)
// is parity(hi & PH[j]) ^ parity(lo & PL[j]) for 112-bit frames and parity(hi & PS[j]) for
// 56-bit frames (CRC-24 is GF(2)-linear in the frame bits).
struct ParityMasks { uint64_t PH[24], PL[24], PS[24]; };
ParityMasks build_parity_masks();

// Syndrome contribution of each 5-bit group of frame bits: the slicer produces the frame five bits
// at a time (bits 5g..5g+4, first bit = MSB of the group value), and CRC-24 being GF(2)-linear the
// syndrome is the XOR over groups of table[g][value].  Bits past the end of the frame contribute 0.
// Layout: long[23][32] followed by short[12][32].
constexpr int kGroupsLong = 23, kGroupsShort = 12;
std::vector<uint32_t> build_group_syndromes();

// UC8 magnitude table, 65536 entries, index = I | Q<<8 (init_uc8_lookup, convert.c:35-62)
const uint16_t *uc8_table();
// The same table folded by its two mirror symmetries: entry [a*UC8_FOLD_STRIDE + b] is the
// magnitude for |I-127.5| = a+0.5, |Q-127.5| = b+0.5 (a,b in 0..127).  33 KB, lives in LDS.
constexpr int UC8_FOLD_STRIDE = 130;   // row stride chosen so a column walk changes LDS bank
// ... followed (at UC8_SYM_OFFSET halfwords) by the same quadrant with row stride 128: 32 KB, k_convert_uc8_lean's form
constexpr int UC8_SYM_OFFSET = 128 * UC8_FOLD_STRIDE;
std::vector<uint16_t> uc8_folded_table();

// tan(roll) for the 1024 roll codes of BDS5,0 (index = sign << 9 | 9-bit magnitude, roll = magnitude * 45/256 - 90 * sign as
// the float comm_b.c:541-545 forms), evaluated as comm_b.c:632 does — tan(roll * M_PI / 180.0) in the host's libm, the same
// library the reference links — so the field decoder's turn-rate consistency test compares the reference's own doubles.
std::vector<double> build_roll_tangent_table();

}  // namespace mgpu
