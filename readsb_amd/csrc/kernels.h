// kernels.h — device data layout and kernel launchers of the Mode-S hot path (gfx950).
//
// HBM layout of one feed of n new samples (one context = one stream):
//   d_iq   : raw IQ, n * {2,4} bytes (UC8 / SC16, SC16Q11)
//   d_mag  : u16 magnitudes, the reference's "delayed stream": d_mag[0..326) = the 326
//            samples that preceded this feed (zeros at stream start), d_mag[326 + i] =
//            magnitude of new sample i.  Index D of d_mag is exactly the scan position
//            `pa - m` of demodulate2400 summed over the 131072-sample buffers
//            (demod_2400.c:287-290, sdr_ifile.c:209-213), so D in [0, n) are the preamble
//            start positions and a position reads d_mag[D .. D+289].
//   pool   : PhaseRec[], per-tile chains of segments written by k_slice
//   live   : PhaseRec[], records surviving the pre-screen, globally ordered by (pos, phase)
#pragma once
#ifndef MGPU_EXPERIMENTS
#define MGPU_EXPERIMENTS 0
#endif
#include <hip/hip_runtime.h>
#include <cstdint>

#include "../../include/modes_gpu.h"   // struct mgpu_msg (the beast encoder reads it on the device)

namespace mgpu {

constexpr int kWinPartWords = 8192 * 8;  // k_window_stats: one row of 8 words per workgroup (kWinMaxBlocks rows)
constexpr int kTrailing = 326;          // Modes.trailing_samples (readsb.c:288)
constexpr int kTile = 4096;             // scan positions per LDS tile
constexpr int kHalo = 304;              // >= 290 samples of look-ahead (demod reads pa[0..289]), multiple of 8
constexpr int kTilesPerUnit = 2;
constexpr int kUnit = kTile * kTilesPerUnit;   // positions per unit (one record chain per unit, one wave's share)
constexpr int kBlock = 256;
// k_sweep: positions per candidate list = one of its pre-check steps (16 positions per lane); k_slice's tiles are 2048 = two lists
constexpr int kSweepTile = 1024;
#ifndef MGPU_SL_WGS_PER_CU
#define MGPU_SL_WGS_PER_CU 4
#endif
constexpr int kSweepMaxBlocks = 256 * MGPU_SL_WGS_PER_CU;  // resident workgroups of k_slice (at most 4 per CU)
constexpr int kSweepGridMax = 256 * 8;    // resident workgroups of k_sweep (at most 8 per CU) = rows of sweep_part
constexpr int kDealerCounters = 64;       // k_slice's tile dealer: pools of workgroups, one counter each ...
constexpr int kDealerStride = 64;         // ... 256 bytes apart: device atomics on words of one 64-byte line serialise with each other (tools/micro/atomic_cost.hip)
constexpr int kPoolChunkRecords = 256;    // pool records a wave reserves per returning atomic
constexpr int kSweepMaxWaves = kSweepMaxBlocks * (kBlock / 64);
constexpr uint32_t kNone = 0xFFFFFFFFu;

// flags of a PhaseRec
enum : uint8_t {
    REC_ACCEPT_IF_UNKNOWN = 1,   // decodeModesMessage accepts it even when the address is not in the ICAO filter
    REC_ADDER = 2,               // a clean DF17 / DF11 IID 0: accepted => icaoFilterAdd(addr) (mode_s.c:766-779)
    REC_DFFIX = 4,               // DF repaired to 17 (fixDF17msgtype, mode_s.c:276-301); fixbit0 = DF bit index
    REC_LONG = 8,                // 14 bytes were sliced (else 7)
    REC_CORR_SHIFT = 4,          // bits 4-5: mm->correctedbits decodeModesMessage will report
    REC_COND = 64,               // score_unknown < 0: only alive if the address is (or may become) known
};

// One scored (candidate position, phase) pair whose score is not -2: the filter-independent
// part of score_phase()+scoreModesMessage() (demod_2400.c:215-258, mode_s.c:309-419).
// A record with phase == 0xFF is a segment header: pos = record count, addr = index of the
// next segment header of the same unit (kNone = last).
struct PhaseRec {
    uint32_t pos;            // scan position D within the feed
    uint8_t phase;           // try-phase 4..8
    uint8_t flags;
    int16_t score_known;     // score if `addr` is in the ICAO filter
    int16_t score_unknown;   // score if it is not (-1 = rejected as unknown ICAO)
    uint8_t fixbit0, fixbit1;  // frame bits decodeModesMessage flips (0xFF = none)
    uint32_t addr;           // address the filter is asked about (AA after repair, or the AP syndrome)
    uint8_t msg[14];         // frame as sliced
    uint16_t pad;
};
static_assert(sizeof(PhaseRec) == 32, "PhaseRec must be 32 bytes");
static_assert(sizeof(mgpu_msg) == 64 && sizeof(mgpu_fields) == 176, "C-ABI record sizes");

// counters produced on the device (indices into a u64 array)
enum {
    CNT_CANDIDATES = 0,      // positions with >= 1 phase tried, before any skip-ahead
    CNT_PHASE0 = 1,          // .. CNT_PHASE0+4: score_phase calls per try-phase 4..8, before skip-ahead
    CNT_RECORDS = 6,
    CNT_POOL_OVERFLOW = 7,
    CNT_CLASS_COND = 8,      // candidates whose records are all conditional (REC_COND)
    CNT_CLASS_UNCOND = 9,    // candidates with >= 1 unconditional record
    CNT_DEBUG0 = 16,         // .. 30: the experiments build's checking kernels count what they print here
    CNT_LIVE_TOTAL = 31,     // records surviving the pre-screen (written by k_scan_units)
    CNT_NUM = 32,
};

struct SweepParams {
    const uint16_t *mag;      // d_mag
    uint64_t n;               // number of scan positions (= new samples)
    int32_t thr;              // preamble threshold (demod_2400.c:335-338)
    uint32_t valid_long;      // valid_df_long_bitset  (demod_2400.c:112-128)
    uint32_t valid_short;     // valid_df_short_bitset
    int32_t fix_df;           // Modes.fixDF && Modes.nfix_crc
    const uint32_t *bit_syndrome;   // [112]
    const uint64_t *parity;         // PH[24], PL[24], PS[24]
    const uint32_t *group_syndrome;  // [23][32] long + [12][32] short (tables.h build_group_syndromes)
    const uint64_t *tab_long;       // packed syndrome table for 112-bit frames
    const uint64_t *tab_short;      // packed syndrome table for 56-bit frames
    int32_t n_long, n_short;
    PhaseRec *pool;
    uint32_t pool_cap;
    uint32_t *pool_used;      // device counter (records incl. headers)
    uint32_t *unit_first;     // k_slice: [tiles of 2048 positions] index of the tile's first segment header, kNone if empty
    uint32_t *unit_count;     // k_slice: [tiles] records in the tile's first segment
    uint32_t *dealer;         // k_slice: [kDealerCounters] tiles dealt from each pool so far, kDealerStride words apart (zero at launch); behind them k_sweep: [kDealerCounters] blocks of steps dealt
    uint32_t nunits;
    uint16_t *cand;           // candidate codes (position in the unit << 3 | phase mask), one list per step of kSweepTile positions, kSweepTile slots each
    uint32_t *cand_count;     // [steps], + one empty list behind an odd number of steps
    uint32_t pace_recip;      // k_sweep: 2^32 / reference step time in 10 ns ticks (set by launch_sweep; 0 = no pacing)
    uint32_t *sweep_part;     // [k_slice workgroups][8] partial counters (records, candidates, phases 4/5, 6/7, 8), summed by the pre-screen write pass
    uint32_t *adder_bitmap;   // 2^24 bits: addresses some clean DF17 / DF11 IID 0 frame carries
    unsigned long long *counters;   // [CNT_NUM]
    // k_sweep_uc8 (converter and sweep in one; iq == nullptr: k_sweep over the magnitudes in `mag`): the chunk's UC8 samples, the 326
    // magnitudes before the chunk (nullptr: zeros), the 128 x 128 symmetric table (tables.h: UC8_SYM_OFFSET), where the magnitudes
    // go (= mag), the buffers' exact sum(mag) / sum(mag^2) the kernel adds its samples' to (zero at launch), steps per sample buffer
    const uint8_t *iq;
    uint32_t iq_format;       // of `iq`: MGPU_FMT_UC8 (k_sweep_uc8) / SC16 / SC16Q11 (k_sweep_sc16<15 / 11>: 4 bytes per sample, no table, no integer sums)
    const uint16_t *tail;
    const uint16_t *uc8_sym;
    uint16_t *mag_w;
    unsigned long long *sum_level, *sum_power;
    uint32_t buf_steps;
#if MGPU_EXPERIMENTS
    int32_t debug_stage;      // k_slice with one part left out (timing experiments, MGPU_DEBUG_STAGE, tools/slice_stages.sh)
    unsigned long long *dbg_waves;   // k_sweep: [waves][2] start / end of every wave, 100 MHz (tools/micro/sweep_cold.hip), or null
#endif
};

struct ConvertParams {
    const uint8_t *iq;
    uint16_t *mag;            // d_mag (written from index kTrailing on)
    uint64_t n;
    uint32_t buf_samples;
    const uint16_t *tail;           // 326 magnitudes preceding the chunk (device), nullptr = zeros
    const uint16_t *uc8_folded;     // device copy of the folded UC8 table
    unsigned long long *sum_level;  // [nbuffers] UC8: exact integer sum of mag
    unsigned long long *sum_power;  // [nbuffers] UC8: exact integer sum of mag^2
    double *fsum_level;             // [nbuffers] SC16*: sum of mag (0..1), double accumulation
    double *fsum_power;             // [nbuffers] SC16*: sum of magsq
};

void launch_convert(int format, const ConvertParams &p, hipStream_t s, unsigned max_blocks = 0, int variant = 0);   // variant 1: the round-1..5 UC8 converter (k_convert_uc8), else k_convert_uc8_lean where the buffer geometry allows
void launch_spin(unsigned us, unsigned blocks, unsigned long long *sink, hipStream_t s);   // a kernel that lasts `us` microseconds (event calibration)
// SC16 formats: the per-buffer sequential float sums of mag / magsq (convert.c:225-249), exact; state in and out as doubles holding floats
void launch_fsum_sc16(int format, const uint8_t *iq, uint64_t n, uint32_t buf_samples, double *fsum_level, double *fsum_power, int want_level,
                      hipStream_t s, int fresh = 0);   // fresh: the sums start at zero (else they continue what fsum_* holds)
#if MGPU_EXPERIMENTS
// the same sums for the pipeline's chunks, wide: approximate block sums from the magnitudes -> per-block summaries against predicted
// binades -> a short apply chain per buffer (kernels/convert.inc); scratch = fsum_wide_scratch_bytes()
size_t fsum_wide_scratch_bytes(uint64_t max_samples, uint32_t buf_samples);
void launch_fsum_sc16_wide(int format, const uint8_t *iq, const uint16_t *mag, uint64_t n, uint32_t buf_samples, double *fsum_level, double *fsum_power,
                           int want_level, void *scratch, hipStream_t s);
#endif
unsigned launch_sweep(const SweepParams &p, hipStream_t s);        // k_sweep: preamble sweep -> per-step candidate lists; returns its grid size
void sweep_pace_feedback(float kernel_us, uint64_t n, unsigned blocks, float bracket_us, int fused = 0);   // a timed k_sweep launch: feeds the pacing's step-time estimate
unsigned launch_slice(const SweepParams &p, hipStream_t s, unsigned max_blocks = 0);        // k_slice: slicer + CRC + scoring over the candidate lists -> record pool; returns its grid size (rows of sweep_part)
// pre-screen: count / write the records whose address may matter to the ordered walk
// (the write pass also stores each live record's would-be signal power: sum of mag^2 over its frame)
// everything between the sweep and the host: class planes -> class bitmap (+ counters, planes zeroed again),
// pre-screen count / scan / write, scratch block to the host (and zeroed again)
struct PostSweepParams {
    PhaseRec *pool;                       // (the count pass leaves live masks and output offsets in the segment headers)
    uint32_t pool_cap;                    // records the pool holds
    uint32_t variant;                     // bit 1: the write pass by chains (write_unit_chains; else the older one, one chain after the other); bit 2 (experiments build): checking kernels; 3 = the product
    const uint32_t *unit_first;           // first segment header of every chain, chains_per_unit consecutive chains per unit
    const uint32_t *first_count;          // four chains per unit: the records in every chain's first segment (SweepParams::unit_count)
    uint32_t chains_per_unit;             // 4: one chain per k_slice tile of 2048 positions
    uint32_t nunits;
    const uint32_t *adder_bitmap;
    uint32_t *unit_live;                  // live records per unit (count pass)
    uint32_t *block_live;                 // ... per count-pass workgroup (4 units)
    PhaseRec *live;                       // pinned host memory
    const uint16_t *mag;
    unsigned long long *live_sig;         // pinned host memory
    unsigned long long *counters;
    uint32_t *class_final;                // the class bitmap (1 bit per scan position: records, all of them conditional), written by the count pass: kUnit / 32 words per unit
    uint64_t class_words;
    uint32_t *dealer;                     // k_slice's dealer counters: zeroed again by k_publish
    const uint16_t *cand; const uint32_t *cand_count;   // k_sweep's candidate lists of the chunk (the live records' windows are counted from them)
    unsigned long long *live_win;         // shard passes: per live record the packed counts of its would-be skip window (null: not wanted)
    uint64_t n; int32_t thr; uint32_t buf_len;   // ... and what that kernel needs: the chunk's positions, the threshold, the buffer length
    unsigned long long *d_scratch, *h_scratch;
    uint32_t scratch_words;
    uint32_t *fin_part;                   // [count-pass workgroups][2] class counts, summed by the write pass
    const uint32_t *slice_part;           // k_slice's rows of counts (SweepParams::sweep_part), summed by the write pass; slice_blocks = 0: none
    uint32_t slice_blocks;
};
int launch_prescreen(const PostSweepParams &q, hipStream_t s, hipStream_t s_write, hipEvent_t ev_scan);
// ---- Mode A/C (demodulate2400AC, demod_2400.c:575-761), only when mgpu_config.mode_ac is set ----
// A position that passes every test of the reference's loop body: the reply is accepted unless an earlier accepted
// reply of the same buffer hides it (f1_sample += 69, :765) — decided in order on the host.
constexpr int kAcLists = 64;   // k_modeac appends to one of 64 lists (own counter word, own slice of the output)
struct AcCand {
    uint32_t pos;        // scan position D in the chunk (f1_sample = D - buffer start)
    uint32_t f2_clock;   // 60 MHz cycles from the buffer start to the F2 pulse: timestamp = sampleTimestamp + f2_clock / 5 (:755)
    uint32_t modeac;     // 00 A4 A2 A1  00 B4 B2 B1  SPI C4 C2 C1  00 D4 D2 D1 (:731-744)
};
void launch_modeac(const uint16_t *mag, uint64_t n, uint32_t buf_samples, int format, const unsigned long long *sum_level,
                   const unsigned long long *sum_power, const double *fsum_level, const double *fsum_power,
                   uint32_t *noise_level, AcCand *out, uint32_t cap, unsigned long long *list_counts /* [kAcLists], zero */,
                   unsigned long long *counters, hipStream_t s);
// beast wire frames (modesSendBeastOutput, net_io.c:1655-1714) of n message records in device memory; len [n] and
// block_bytes [ceil(n/256)] are scratch; *total (device) receives the number of bytes the frames take
void launch_modeac_scan(const uint16_t *mag, uint64_t n, uint32_t buf_samples, const uint32_t *noise_level, AcCand *out, uint32_t cap,
                        unsigned long long *list_counts, unsigned long long *counters, hipStream_t s);
void launch_decode_fields(const mgpu_msg *msgs, uint64_t n, mgpu_fields *out, const double *roll_tan, hipStream_t s);
void launch_beast_encode(const mgpu_msg *msgs, uint64_t n, uint16_t *meta, uint32_t *block_bytes, unsigned long long *block_off, uint8_t *out,
                         uint64_t cap, unsigned long long *total /* [2]: bytes, deferred */, hipStream_t s, const uint8_t *verdict = nullptr, int net_rule = 0,
                         uint32_t *block_def = nullptr, unsigned long long *block_def_off = nullptr, mgpu_deferred *deferred = nullptr, uint64_t def_cap = 0);
// first stage of the tracker + the forwarding rule over a message list in device memory (kernels/gate.inc): table = gate_table_bytes()
// bytes, zeroed once and kept from call to call; scratch = gate_scratch_bytes(n); verdict: one byte per message (include/modes_gpu.h)
size_t gate_table_bytes();
size_t gate_scratch_bytes(uint64_t n);
void launch_track_gate(const mgpu_msg *msgs, const mgpu_fields *fields, uint64_t n, uint32_t buf_samples, void *table, void *scratch,
                       uint8_t *verdict, hipStream_t s);
// signal power of accepted messages: sum of mag^2 over d_mag[pos+19 .. pos+19+len)
void launch_signal_power(const uint16_t *mag, const uint32_t *pos, const uint16_t *len, uint32_t nmsg,
                         unsigned long long *out, hipStream_t s);
// candidates / tried phases / conditional-class candidates inside the skip-ahead window of
// each accepted message (positions pos+1 .. pos+skip, clipped to `limit`), for the stats fix-up
// struct modesMessage fields of the accepted frames on the device (kernels/build.inc): acc = Accepted[nacc], bufs = BufferClock[]
void launch_build_messages(const PhaseRec *live, const unsigned long long *live_sig, const unsigned long long *msg_sig, const void *d_acc,
                           const void *d_bufs, uint32_t nacc, mgpu_msg *out, hipStream_t s);   // msg_sig (per accepted frame) or, when null, live_sig (per live record)
void launch_stage_blob(const void *h_src, void *d_dst, uint64_t bytes, hipStream_t s);   // page-locked host -> device, small grid

// ---- the ordered accept walk on the device (kernels/walk.inc) ----
constexpr uint32_t kWkNoAdd = 0xffffffffu;
constexpr uint32_t kWkTouchedCap = 1u << 16;        // distinct adder addresses per chunk and table
constexpr int kWkCounts = 16;                       // per-buffer counters (WKC_* in kernels/walk.inc, the order of ResolveCounts)
constexpr int32_t kWkNoFlip = 0x7fffffff;
struct WalkIn {                  // head of the per-chunk input blob; then (at kWkInHead) BufferClock[nbuf], active[n_active], inactive[n_inactive]
    int64_t next_flip;           // Resolver::next_flip(): the expiry is due at the first buffer end with clock >= this
    uint32_t nbuf, n_active, n_inactive, acc_cap;   // acc_cap: accept-list capacity per buffer
    uint32_t nlive, buf_len, pad[2];   // live records of the chunk (the device's own counter block is back to zero by now); buffer b = positions [b * buf_len, ..)
};
static_assert(sizeof(WalkIn) == 40, "WalkIn");
constexpr uint32_t kWkInHead = 48;                  // the buffer clocks start here
struct WalkState {
    int32_t flip;                // the expiry is assumed after this buffer (kWkNoFlip: not in this chunk)
    uint32_t converged, iterations, bad;
    uint32_t n_touched[2];
    uint32_t cur, done;          // done: workgroups of the running kernel that have finished (last one does the serial part)
};
struct WalkBuffers {             // device memory of the walk (one set per context)
    uint32_t *bit_active, *bit_inactive;             // 2^24 bits each: the filter's two generations when the chunk starts
    uint32_t *first[2];                              // 2^24 entries each: first buffer of the chunk that adds the address
    uint32_t *touched[2];                            // addresses with an entry in first[t]
    WalkState *state;
    uint32_t *rec_lo;                                // [nbuf + 1]: first live record at or after the buffer's first position
    uint32_t *nacc, *nadds, *counts;                 // per buffer
    long long *end_clock;                            // per buffer: Modes.synthetic_now at its end
    uint32_t *acc;                                   // per buffer acc_cap x (record index, score)
    uint32_t *adds;                                  // per buffer acc_cap addresses, in order of their first add in the buffer
    uint32_t *offs, *aoffs;                          // [nbuf + 1] prefix of nacc / of nadds
};
struct WalkSummary {             // what the host reads (page-locked): this, then per buffer 6 words {accepted, adds, end clock lo, hi,
    uint32_t converged, bad, iterations, nmsg;       // offset into the adds, offset into the accept list}, then the adds
    int32_t flip; uint32_t nadds_total, nbuf, pad;
    unsigned long long counts[kWkCounts];
};
size_t walk_summary_bytes(uint32_t nbuf, uint32_t acc_cap);
size_t walk_input_bytes(uint32_t nbuf, uint32_t n_active, uint32_t n_inactive);
void launch_device_walk(const uint8_t *h_in, uint8_t *d_in, size_t in_bytes, const WalkBuffers &w, const PhaseRec *live, uint32_t nbuf,
                        void *h_summary, void *d_acc, uint32_t *msg_pos, uint32_t *msg_limit, uint16_t *msg_skip, uint32_t msg_cap, hipStream_t s);
// signal power of the accepted frames after the walk: out[i] = sum of mag^2 over frame i | 1 << 63 for a 112-bit frame
void launch_msg_sig(const uint16_t *mag, const uint32_t *d_pos, const uint16_t *d_skip, uint32_t n, unsigned long long *d_out, unsigned long long *h_out,
                    hipStream_t s);   // h_out (page-locked host memory, may be null): a second copy for the builder
void launch_stage_in(const uint32_t *h_pos, const uint32_t *h_limit, const uint16_t *h_skip, uint32_t *d_pos, uint32_t *d_limit,
                     uint16_t *d_skip, uint32_t n, hipStream_t s);   // page-locked host arrays -> device, small grid
// shard passes: the same numbers for every live record's would-be window, packed into out[record] (kernels/window_stats.inc)
void launch_live_windows(const uint16_t *mag, uint64_t n, const uint16_t *cand, const uint32_t *cand_count, const uint32_t *class_bitmap, const PhaseRec *live, const unsigned long long *nlive_dev,
                         uint32_t buf_len, unsigned long long *out, hipStream_t s);
void launch_window_stats(const uint16_t *mag, uint64_t n, const uint16_t *cand, const uint32_t *cand_count, const uint32_t *class_bitmap, const uint32_t *pos,   // cand / cand_count: k_sweep's lists of the chunk
                         const uint16_t *skip, const uint32_t *limit, uint32_t nmsg, unsigned long long *part, unsigned long long *out,
                         hipStream_t s, unsigned long long *sig_out = nullptr, unsigned long long *sig_host = nullptr,
                         hipEvent_t ev_sig = nullptr);   // sig_out: also the frames' signal powers (k_msg_sig's form), sig_host: a second copy in page-locked host memory, ev_sig: recorded behind them

}  // namespace mgpu
