// tables.cpp — host-side construction of the constant tables (see tables.h).
#include "tables.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <mutex>

namespace mgpu {

static constexpr uint32_t kPoly = 0xfff409u;   // crc.c:31

CrcTables::CrcTables() {
    for (uint32_t v = 0; v < 256; ++v) {
        uint32_t r = v << 16;
        for (int s = 0; s < 8; ++s) r = (r & 0x800000u) ? ((r << 1) ^ kPoly) : (r << 1);
        byte_table[v] = r & 0xffffffu;
    }
    uint8_t frame[14];
    for (int k = 0; k < 112; ++k) {
        std::memset(frame, 0, sizeof(frame));
        frame[k >> 3] = uint8_t(0x80u >> (k & 7));
        bit_syndrome[k] = checksum(frame, 112);
    }
}

uint32_t CrcTables::checksum(const uint8_t *msg, int bits) const {
    const int nbytes = bits / 8;
    uint32_t rem = 0;
    for (int i = 0; i < nbytes - 3; ++i)
        rem = ((rem << 8) ^ byte_table[msg[i] ^ ((rem >> 16) & 0xff)]) & 0xffffffu;
    return rem ^ (uint32_t(msg[nbytes - 3]) << 16) ^ (uint32_t(msg[nbytes - 2]) << 8) ^ msg[nbytes - 1];
}

const CrcTables &crc_tables() {
    static const CrcTables t;
    return t;
}

std::vector<SyndromeEntry> build_syndrome_table(int bits, int nfix) {
    std::vector<SyndromeEntry> tab;
    if (nfix <= 0) return tab;
    const CrcTables &crc = crc_tables();
    const int off = 112 - bits;
    const int max_correct = nfix >= 2 ? 2 : 1;
    const int max_detect = nfix >= 2 ? 4 : 1;
    auto syn = [&](int k) { return crc.bit_syndrome[k + off]; };

    // every 1..max_correct-bit pattern over frame bits 5..bits-1 (the DF field is excluded, crc.c:211)
    for (int a = 5; a < bits; ++a) {
        tab.push_back({syn(a), 1, int8_t(a), -1});
        if (max_correct >= 2)
            for (int b = a + 1; b < bits; ++b) tab.push_back({syn(a) ^ syn(b), 2, int8_t(a), int8_t(b)});
    }
    std::stable_sort(tab.begin(), tab.end(),
                     [](const SyndromeEntry &x, const SyndromeEntry &y) { return x.syndrome < y.syndrome; });
    // a syndrome reachable by two different patterns is ambiguous: drop all of them (crc.c:232-249)
    {
        std::vector<SyndromeEntry> uniq;
        for (size_t i = 0; i < tab.size();) {
            size_t j = i + 1;
            while (j < tab.size() && tab[j].syndrome == tab[i].syndrome) ++j;
            if (j == i + 1) uniq.push_back(tab[i]);
            i = j;
        }
        tab.swap(uniq);
    }
    // drop entries that a (max_correct+1 .. max_detect)-bit pattern would also produce (crc.c:252-283)
    if (max_detect > max_correct) {
        std::vector<uint8_t> dead(tab.size(), 0);
        auto mark = [&](uint32_t s) {
            auto it = std::lower_bound(tab.begin(), tab.end(), s,
                                       [](const SyndromeEntry &e, uint32_t v) { return e.syndrome < v; });
            if (it != tab.end() && it->syndrome == s) dead[it - tab.begin()] = 1;
        };
        for (int a = 5; a < bits; ++a)
            for (int b = a + 1; b < bits; ++b)
                for (int c = b + 1; c < bits; ++c) {
                    const uint32_t s3 = syn(a) ^ syn(b) ^ syn(c);
                    mark(s3);
                    for (int d = c + 1; d < bits; ++d) mark(s3 ^ syn(d));
                }
        std::vector<SyndromeEntry> kept;
        for (size_t i = 0; i < tab.size(); ++i)
            if (!dead[i]) kept.push_back(tab[i]);
        tab.swap(kept);
    }
    return tab;
}

std::vector<uint64_t> pack_syndrome_table(const std::vector<SyndromeEntry> &t) {
    std::vector<uint64_t> out;
    out.reserve(t.size());
    for (const auto &e : t)
        out.push_back((uint64_t(e.syndrome) << 16) | (uint64_t(uint8_t(e.bit0)) << 8) |
                      uint64_t(e.nerr >= 2 ? uint8_t(e.bit1) : 0xFFu));
    return out;
}

ParityMasks build_parity_masks() {
    const CrcTables &crc = crc_tables();
    ParityMasks m;
    std::memset(&m, 0, sizeof(m));
    for (int j = 0; j < 24; ++j) {
        for (int k = 0; k < 64; ++k)
            if ((crc.bit_syndrome[k] >> j) & 1) m.PH[j] |= 1ull << (63 - k);
        for (int k = 64; k < 112; ++k)
            if ((crc.bit_syndrome[k] >> j) & 1) m.PL[j] |= 1ull << (47 - (k - 64));
        for (int k = 0; k < 56; ++k)
            if ((crc.bit_syndrome[k + 56] >> j) & 1) m.PS[j] |= 1ull << (63 - k);
    }
    return m;
}

std::vector<uint32_t> build_group_syndromes() {
    const CrcTables &crc = crc_tables();
    std::vector<uint32_t> t((kGroupsLong + kGroupsShort) * 32, 0);
    for (int g = 0; g < kGroupsLong; ++g)
        for (int v = 0; v < 32; ++v) {
            uint32_t s = 0;
            for (int i = 0; i < 5; ++i) {
                const int k = 5 * g + i;
                if (k < 112 && ((v >> (4 - i)) & 1)) s ^= crc.bit_syndrome[k];
            }
            t[g * 32 + v] = s;
        }
    for (int g = 0; g < kGroupsShort; ++g)
        for (int v = 0; v < 32; ++v) {
            uint32_t s = 0;
            for (int i = 0; i < 5; ++i) {
                const int k = 5 * g + i;
                if (k < 56 && ((v >> (4 - i)) & 1)) s ^= crc.bit_syndrome[k + 56];
            }
            t[(kGroupsLong + g) * 32 + v] = s;
        }
    return t;
}

const uint16_t *uc8_table() {
    static std::vector<uint16_t> tab;
    static std::once_flag once;
    std::call_once(once, [] {
        tab.resize(65536);
        // convert.c:45-58: fI, fQ are the double quotients rounded to float; the rest is float
        // arithmetic with one rounding per operation (no fused multiply-add on the reference's
        // x86-64 build), sqrtf is correctly rounded.
        for (int i = 0; i < 256; ++i) {
            const float fI = float((i - 127.5) / 127.5);
            const volatile float fI2 = fI * fI;
            for (int q = 0; q < 256; ++q) {
                const float fQ = float((q - 127.5) / 127.5);
                const volatile float fQ2 = fQ * fQ;
                volatile float magsq = fI2 + fQ2;
                if (magsq > 1.0f) magsq = 1.0f;
                const volatile float mag = sqrtf(magsq);
                const volatile float scaled = mag * 65535.0f;
                const volatile float rounded = scaled + 0.5f;
                tab[i * 256 + q] = uint16_t(rounded);   // == lut[le16 pair I|Q<<8] by symmetry
            }
        }
    });
    return tab.data();
}

std::vector<uint16_t> uc8_folded_table() {
    const uint16_t *full = uc8_table();
    std::vector<uint16_t> f(128 * UC8_FOLD_STRIDE + 128 * 128, 0);
    for (int a = 0; a < 128; ++a)
        for (int b = 0; b < 128; ++b) f[a * UC8_FOLD_STRIDE + b] = full[(128 + a) * 256 + (128 + b)];
    // behind it the same quadrant without the padding (k_convert_uc8_lean: the byte address of an entry is the folded sample
    // pair itself, high byte << 8 | low byte << 1 — the quadrant is symmetric, so which of I and Q is the row does not matter)
    for (int a = 0; a < 128; ++a)
        for (int b = 0; b < 128; ++b) f[128 * UC8_FOLD_STRIDE + a * 128 + b] = full[(128 + a) * 256 + (128 + b)];
    return f;
}

std::vector<double> build_roll_tangent_table() {
    std::vector<double> t(1024);
    for (unsigned code = 0; code < 1024; ++code) {
        float roll = (float) ((code & 511u) * 45.0 / 256.0);
        if (code & 512u) roll = (float) (roll - 90.0);
        t[code] = std::tan(roll * M_PI / 180.0);
    }
    return t;
}

}  // namespace mgpu
