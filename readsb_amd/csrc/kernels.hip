// kernels.hip — hand-written HIP kernels (gfx950 / CDNA4, wave64) for readsb's Mode-S hot path.
//
//   k_convert_*      IQ -> u16 magnitude                   (convert.c:64-108, 212-250, 329-367)
//   k_sweep_slice    preamble sweep over every sample position, wave-ballot/prefix compaction of
//                    the candidates, PPM bit slicing of each tried phase straight from the
//                    LDS-staged sample window, wave-parallel CRC-24, syndrome lookup and the
//                    filter-independent half of scoreModesMessage
//                                                          (demod_2400.c:74-93,133-258,290-378;
//                                                           mode_s.c:276-419; crc.c:67-82,383-406)
//   k_prescreen_*    drops records that can only ever score "unknown ICAO"
//   k_signal_power   sum of mag^2 over an accepted frame  (demod_2400.c:436-457)
//   k_window_stats   what the skip-ahead hid from the counters (demod_2400.c:468)
//
// No MFMA anywhere: this is HBM-bound integer/byte streaming work.  All arithmetic on the
// message path is integer and bit-exact with the reference; the SC16 converters use IEEE float
// ops with contraction disabled and a correctly rounded sqrt.
#include "kernels.h"
#include "tables.h"

#include <cstdlib>

namespace mgpu {

#define WAVE 64

// Per-stage cycle counters of k_sweep_slice (thread 0 of every workgroup, summed into
// counters[10..22], printed by api.cpp when MGPU_DEBUG_PRINT is set).  Off in production builds:
// every s_memtime is a scalar-memory round trip.
#ifndef MGPU_KERNEL_TIMERS
#define MGPU_KERNEL_TIMERS 0
#endif
#if MGPU_KERNEL_TIMERS
#define DBG_CLOCK() clock64()
#else
#define DBG_CLOCK() 0ll
#endif

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
typedef uint32_t u32x4_a8 __attribute__((ext_vector_type(4), aligned(8)));
typedef unsigned short v2u16 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int lane_id() { return threadIdx.x & (WAVE - 1); }

__device__ __forceinline__ uint32_t rfl(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }

__device__ __forceinline__ uint64_t readlane64(uint64_t v, int src) {
    uint32_t lo = __builtin_amdgcn_readlane((uint32_t) v, src);
    uint32_t hi = __builtin_amdgcn_readlane((uint32_t) (v >> 32), src);
    return ((uint64_t) hi << 32) | lo;
}

// inclusive->exclusive wave prefix sum; total = sum over the wave
__device__ __forceinline__ int wave_excl_scan(int v, int &total) {
    const int lane = lane_id();
    int x = v;
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
        int y = __shfl_up(x, d);
        if (lane >= d) x += y;
    }
    total = __shfl(x, WAVE - 1);
    return x - v;
}

// Exclusive prefix sum over the wave of a small per-lane count (< 16), bit-sliced: four ballots and
// v_mbcnt — no cross-lane data movement, where __shfl_up costs a trip through the LDS crossbar per step.
__device__ __forceinline__ int wave_excl_scan_small(uint32_t v, int &total) {
    int ex = 0;
    total = 0;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const uint64_t m = __ballot((v >> b) & 1u);
        ex += (int) __builtin_amdgcn_mbcnt_hi((uint32_t) (m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) m, 0u)) << b;
        total += __popcll(m) << b;
    }
    return ex;
}

__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    return v;
}

// =============================================================================================
// IQ -> magnitude
// =============================================================================================

// Per-thread running (level, power) sums for the buffer the thread is currently inside.  A thread
// that crosses a 131072-sample buffer boundary, or ends, folds its sums into workgroup accumulators
// in LDS (a workgroup's contiguous range touches kBlockBufs buffers at most in the usual geometry;
// anything beyond goes to memory directly); one thread per touched buffer then issues the
// device-scope atomics.  One RMW per thread was ~5x10^5 memory-side atomics on <= 1024 addresses per
// launch and cost more than the conversion itself.
constexpr int kBlockBufs = 4;
struct BlockSums {
    unsigned long long level[kBlockBufs], power[kBlockBufs];
    double flevel[kBlockBufs], fpower[kBlockBufs];
    uint32_t first;          // buffer index of slot 0
};

struct BufSums {
    unsigned long long level, power;
    uint64_t next_boundary;   // first sample index of the next buffer
    uint32_t cur;             // current buffer index
    BlockSums *blk;
    __device__ void init(uint64_t sample, uint32_t B, BlockSums *b) {
        cur = (uint32_t) (sample / B);
        next_boundary = (uint64_t) (cur + 1) * B;
        level = power = 0;
        blk = b;
    }
    __device__ void flush(const ConvertParams &p) {
        if (level | power) {
            const uint32_t k = cur - blk->first;
            if (k < (uint32_t) kBlockBufs) {
                atomicAdd(&blk->level[k], level);
                atomicAdd(&blk->power[k], power);
            } else {
                atomicAdd(&p.sum_level[cur], level);
                atomicAdd(&p.sum_power[cur], power);
            }
        }
        level = power = 0;
    }
    __device__ void advance_to(uint64_t sample, const ConvertParams &p) {
        while (sample >= next_boundary) {
            flush(p);
            ++cur;
            next_boundary += p.buf_samples;
        }
    }
};

__device__ __forceinline__ void block_sums_init(BlockSums &b, uint64_t first_sample, uint32_t B) {
    if (threadIdx.x < kBlockBufs) {
        b.level[threadIdx.x] = b.power[threadIdx.x] = 0;
        b.flevel[threadIdx.x] = b.fpower[threadIdx.x] = 0.0;
    }
    if (threadIdx.x == 0) b.first = (uint32_t) (first_sample / B);
}

// d_mag[0 .. 326) = the 326 magnitudes that preceded this chunk (sdr_ifile.c:209-213): copied from the
// end of the previous chunk's magnitude buffer (p.tail) or zero at stream start, by workgroup 0.
__device__ __forceinline__ void convert_tail_prologue(const ConvertParams &p) {
    if (blockIdx.x != 0) return;
    for (int i = threadIdx.x; i < kTrailing; i += kBlock) p.mag[i] = p.tail ? p.tail[i] : (uint16_t) 0;
}

// UC8: magnitude = table[I | Q<<8] (convert.c:64-108).  The 65536-entry table has two mirror
// symmetries (I -> 255-I, Q -> 255-Q) so a 128x128 quadrant, padded to an odd-ish row stride,
// is staged in LDS (33 KB) and every sample is one ds_read_u16.
// Thread work item: one 16-byte-aligned chunk of 8 output magnitudes d_mag[8c .. 8c+8) =
// samples 8c-326 .. 8c-319; the 16 IQ bytes are one (4-byte aligned) global_load_dwordx4.
__global__ __launch_bounds__(kBlock) void k_convert_uc8(ConvertParams p) {
    __shared__ __attribute__((aligned(16))) uint16_t s_lut[128 * UC8_FOLD_STRIDE];
    {   // 33 KB table: all of a thread's 16-byte loads are issued before the first LDS store
        constexpr int kVec = 128 * UC8_FOLD_STRIDE / 8;                 // 2080 x 16 B
        constexpr int kPer = (kVec + kBlock - 1) / kBlock;
        u32x4 t[kPer];
#pragma unroll
        for (int k = 0; k < kPer; ++k) { const int i = threadIdx.x + k * kBlock; if (i < kVec) t[k] = ((const u32x4 *) p.uc8_folded)[i]; }
#pragma unroll
        for (int k = 0; k < kPer; ++k) { const int i = threadIdx.x + k * kBlock; if (i < kVec) ((u32x4 *) s_lut)[i] = t[k]; }
    }
    convert_tail_prologue(p);

    const uint64_t c_first = kTrailing / 8;                       // chunk holding d_mag[326]
    const uint64_t c_end = (kTrailing + p.n + 7) / 8;
    const uint64_t nchunks = c_end - c_first;
    uint64_t per_block = (nchunks + gridDim.x - 1) / gridDim.x;
    per_block = (per_block + kBlock - 1) / kBlock * kBlock;
    const uint64_t blk_lo = c_first + (uint64_t) blockIdx.x * per_block;
    const uint64_t blk_hi = blk_lo + per_block < c_end ? blk_lo + per_block : c_end;
    if (blk_lo >= c_end) return;                                  // workgroup-uniform
    __shared__ BlockSums s_sums;
    {
        const int64_t b0 = (int64_t) blk_lo * 8 - kTrailing;
        block_sums_init(s_sums, b0 < 0 ? 0 : (uint64_t) b0, p.buf_samples);
    }
    __syncthreads();

    BufSums sums;
    {
        int64_t s0 = (int64_t) (blk_lo + threadIdx.x) * 8 - kTrailing;
        sums.init(s0 < 0 ? 0 : (uint64_t) s0, p.buf_samples, &s_sums);
    }
    for (uint64_t c = blk_lo + threadIdx.x; c < blk_hi; c += kBlock) {
        const int64_t i0 = (int64_t) c * 8 - kTrailing;   // sample index of element 0
        uint32_t w[4];
        const bool full = i0 >= 0 && (uint64_t) i0 + 8 <= p.n;
        if (full) {
            u32x4_a4 v = *(const u32x4_a4 *) (p.iq + 2 * i0);
            w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
        } else {
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                uint32_t x = 0;
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    int64_t byte = 2 * i0 + 4 * d + b;
                    if (byte >= 0 && (uint64_t) byte < 2 * p.n) x |= (uint32_t) p.iq[byte] << (8 * b);
                }
                w[d] = x;
            }
        }
        uint32_t out[4];
        uint32_t lvl = 0;
        unsigned long long pw = 0;
        uint16_t m[8];
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            // fold all four bytes: v >= 128 ? v - 128 : 127 - v
            const uint32_t hi = (w[d] >> 7) & 0x01010101u;
            const uint32_t f = (w[d] ^ 0x7F7F7F7Fu ^ (hi * 0x7Fu)) & 0x7F7F7F7Fu;
            const uint32_t a0 = f & 0xff, b0 = (f >> 8) & 0xff, a1 = (f >> 16) & 0xff, b1 = f >> 24;
            m[2 * d] = s_lut[a0 * UC8_FOLD_STRIDE + b0];
            m[2 * d + 1] = s_lut[a1 * UC8_FOLD_STRIDE + b1];
            out[d] = (uint32_t) m[2 * d] | ((uint32_t) m[2 * d + 1] << 16);
        }
        if (full) {
            if ((uint64_t) i0 + 7 >= sums.next_boundary || (uint64_t) i0 < sums.next_boundary - p.buf_samples) {
                // chunk straddles (or jumps) a buffer boundary: element-wise
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    sums.advance_to((uint64_t) i0 + e, p);
                    sums.level += m[e];
                    sums.power += (unsigned long long) m[e] * m[e];
                }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) { lvl += m[e]; pw += (unsigned long long) ((uint32_t) m[e] * (uint32_t) m[e]); }
                sums.level += lvl;
                sums.power += pw;
            }
            u32x4 o = {out[0], out[1], out[2], out[3]};
            *(u32x4 *) (p.mag + c * 8) = o;
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int64_t i = i0 + e;
                if (i >= 0 && (uint64_t) i < p.n) {
                    sums.advance_to((uint64_t) i, p);
                    sums.level += m[e];
                    sums.power += (unsigned long long) m[e] * m[e];
                    p.mag[c * 8 + e] = m[e];
                }
            }
        }
    }
    sums.flush(p);
    __syncthreads();
    if (threadIdx.x < kBlockBufs && (s_sums.level[threadIdx.x] | s_sums.power[threadIdx.x])) {
        atomicAdd(&p.sum_level[s_sums.first + threadIdx.x], s_sums.level[threadIdx.x]);
        atomicAdd(&p.sum_power[s_sums.first + threadIdx.x], s_sums.power[threadIdx.x]);
    }
}

// SC16 / SC16Q11: mag = sqrtf(min(1, fI*fI + fQ*fQ)), fI = I/scale, u16 = (uint16_t)(mag*65535+0.5)
// (convert.c:212-250, 329-367).  Every float operation is rounded on its own (the reference is
// scalar SSE built with -std=c11, i.e. no FMA contraction); sqrtf is the correctly rounded one.
template <int SCALE_SHIFT>
__global__ __launch_bounds__(kBlock) void k_convert_sc16(ConvertParams p) {
#pragma clang fp contract(off)
    convert_tail_prologue(p);
    const uint64_t c_first = kTrailing / 8;
    const uint64_t c_end = (kTrailing + p.n + 7) / 8;
    const uint64_t nchunks = c_end - c_first;
    uint64_t per_block = (nchunks + gridDim.x - 1) / gridDim.x;
    per_block = (per_block + kBlock - 1) / kBlock * kBlock;
    const uint64_t blk_lo = c_first + (uint64_t) blockIdx.x * per_block;
    const uint64_t blk_hi = blk_lo + per_block < c_end ? blk_lo + per_block : c_end;
    if (blk_lo >= c_end) return;                                  // workgroup-uniform
    __shared__ BlockSums s_sums;
    {
        const int64_t b0 = (int64_t) blk_lo * 8 - kTrailing;
        block_sums_init(s_sums, b0 < 0 ? 0 : (uint64_t) b0, p.buf_samples);
    }
    __syncthreads();
    const float inv = 1.0f / (float) (1 << SCALE_SHIFT);   // power of two: I * inv == I / scale exactly
    auto fold = [&](uint32_t buf, double l, double w) __attribute__((always_inline)) {
        const uint32_t k = buf - s_sums.first;
        if (k < (uint32_t) kBlockBufs) { atomicAdd(&s_sums.flevel[k], l); atomicAdd(&s_sums.fpower[k], w); }
        else { atomicAdd(&p.fsum_level[buf], l); atomicAdd(&p.fsum_power[buf], w); }
    };

    double lvl = 0.0, pw = 0.0;
    uint32_t cur = 0xFFFFFFFFu;
    for (uint64_t c = blk_lo + threadIdx.x; c < blk_hi; c += kBlock) {
        const int64_t i0 = (int64_t) c * 8 - kTrailing;
        uint16_t m[8];
        const bool full = i0 >= 0 && (uint64_t) i0 + 8 <= p.n;
        uint32_t w[8];
        if (full) {
            u32x4_a8 v0 = *(const u32x4_a8 *) (p.iq + 4 * i0);
            u32x4_a8 v1 = *(const u32x4_a8 *) (p.iq + 4 * i0 + 16);
            w[0] = v0.x; w[1] = v0.y; w[2] = v0.z; w[3] = v0.w;
            w[4] = v1.x; w[5] = v1.y; w[6] = v1.z; w[7] = v1.w;
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int64_t i = i0 + e;
                w[e] = (i >= 0 && (uint64_t) i < p.n) ? *(const uint32_t *) (p.iq + 4 * i) : 0u;
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int64_t i = i0 + e;
            const float fI = (float) (int16_t) (w[e] & 0xffff) * inv;
            const float fQ = (float) (int16_t) (w[e] >> 16) * inv;
            const float a = fI * fI;
            const float b = fQ * fQ;
            float magsq = a + b;
            if (magsq > 1.0f) magsq = 1.0f;
            const float mag = sqrtf(magsq);
            const float sc = mag * 65535.0f;
            const float rd = sc + 0.5f;
            m[e] = (uint16_t) rd;
            if (i >= 0 && (uint64_t) i < p.n) {
                const uint32_t b_e = (uint32_t) ((uint64_t) i / p.buf_samples);
                if (b_e != cur) {
                    if (cur != 0xFFFFFFFFu) fold(cur, lvl, pw);
                    cur = b_e; lvl = pw = 0.0;
                }
                lvl += (double) mag;
                pw += (double) magsq;
            }
        }
        if (full) {
            u32x4 o = {(uint32_t) m[0] | ((uint32_t) m[1] << 16), (uint32_t) m[2] | ((uint32_t) m[3] << 16),
                       (uint32_t) m[4] | ((uint32_t) m[5] << 16), (uint32_t) m[6] | ((uint32_t) m[7] << 16)};
            *(u32x4 *) (p.mag + c * 8) = o;
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int64_t i = i0 + e;
                if (i >= 0 && (uint64_t) i < p.n) p.mag[c * 8 + e] = m[e];
            }
        }
    }
    if (cur != 0xFFFFFFFFu) fold(cur, lvl, pw);
    __syncthreads();
    if (threadIdx.x < kBlockBufs && (s_sums.flevel[threadIdx.x] != 0.0 || s_sums.fpower[threadIdx.x] != 0.0)) {
        atomicAdd(&p.fsum_level[s_sums.first + threadIdx.x], s_sums.flevel[threadIdx.x]);
        atomicAdd(&p.fsum_power[s_sums.first + threadIdx.x], s_sums.fpower[threadIdx.x]);
    }
}

void launch_convert(int format, const ConvertParams &p, hipStream_t s) {
    if (p.n == 0) return;
    const uint64_t nchunks = (kTrailing + p.n + 7) / 8 - kTrailing / 8;
    uint64_t blocks = (nchunks + kBlock * 8 - 1) / (kBlock * 8);   // >= 8 chunks per thread
    if (blocks < 1) blocks = 1;
    if (blocks > 1024) blocks = 1024;                                // 4 x 33 KB LDS per CU
    if (format == 0) hipLaunchKernelGGL(k_convert_uc8, dim3((unsigned) blocks), dim3(kBlock), 0, s, p);
    else if (format == 1) hipLaunchKernelGGL(k_convert_sc16<15>, dim3((unsigned) blocks * 2), dim3(kBlock), 0, s, p);
    else hipLaunchKernelGGL(k_convert_sc16<11>, dim3((unsigned) blocks * 2), dim3(kBlock), 0, s, p);
}

// =============================================================================================
// preamble sweep + bit slicer + CRC/score
// =============================================================================================

// slice_phase0..4 (demod_2400.c:74-93) as rows of a 5x4 table
__constant__ int c_slice_coef[5][4] = {{18, -15, -3, 0}, {14, -5, -9, 0}, {16, 5, -20, 0}, {7, 11, -18, 0}, {4, 15, -20, 1}};

struct LaneConst {
    uint64_t PH, PL, PS;   // CRC parity masks of syndrome bit `lane` (lanes >= 24: zero)
};

// modesChecksumDiagnose (crc.c:383-406) as a wave-cooperative two-level search of the sorted,
// packed table: 64 pivots, then the <=64 entries of the pivot's bucket.  All arguments and
// results are wave-uniform.
__device__ __forceinline__ int wave_diagnose(const uint64_t *tab, int n, uint32_t synd, int &b0, int &b1) {
    if (n <= 0) return -1;
    const int lane = lane_id();
    const int stride = (n + WAVE - 1) / WAVE;
    const int i1 = lane * stride;
    const uint64_t v = i1 < n ? tab[i1] : ~0ull;
    const uint64_t le = __ballot((uint32_t) (v >> 16) <= synd && i1 < n);
    if (le == 0) return -1;
    const int base = (__popcll(le) - 1) * stride;
    const int i2 = base + lane;
    const uint64_t w = (lane < stride && i2 < n) ? tab[i2] : ~0ull;
    const uint64_t hit = __ballot((uint32_t) (w >> 16) == synd && lane < stride && i2 < n);
    if (hit == 0) return -1;
    const uint64_t e = readlane64(w, __ffsll((unsigned long long) hit) - 1);
    b0 = (int) ((e >> 8) & 0xff);
    b1 = (int) (e & 0xff);
    return b1 == 0xff ? 1 : 2;
}

// correct_aa_field (mode_s.c:230-245)
__device__ __forceinline__ uint32_t fix_aa(uint32_t aa, int bit) {
    return (bit >= 8 && bit <= 31) ? aa ^ (1u << (31 - bit)) : aa;
}

// One try-phase of one candidate, executed by a whole wave: slice (lane = frame bit), CRC,
// syndrome lookup, score.  Writes the record into *slot (LDS) and returns its flags | 0x100,
// or returns 0 when the phase scores -2 whatever the ICAO filter holds.
__device__ __forceinline__ uint32_t slice_and_score(const SweepParams &p, const LaneConst &lc, const uint16_t *s_mag,
                                                    const int (*s_coef)[4], int pos_local, uint32_t pos, int t,
                                                    PhaseRec *slot) {
    const int lane = lane_id();
    // ---- bits 0..63 (slice_byte's closed form) ----
    uint64_t hi, lo = 0;
    {
        const int u = (t % 5) + 12 * lane;
        const int q = u / 5, sub = u - 5 * q;
        const uint16_t *s = s_mag + pos_local + 19 + t / 5 + q;
        const int corr = s_coef[sub][0] * s[0] + s_coef[sub][1] * s[1] + s_coef[sub][2] * s[2] + s_coef[sub][3] * s[3];
        hi = __brevll(__ballot(corr > 0));
    }
    const uint32_t df = (uint32_t) (hi >> 59);
    const bool is_long = (p.valid_long >> df) & 1;
    if (!is_long && !((p.valid_short >> df) & 1)) return 0;   // score_phase: invalid DF -> -2
    if (is_long) {
        const int k = 64 + lane;
        const int u = (t % 5) + 12 * k;
        const int q = u / 5, sub = u - 5 * q;
        const uint16_t *s = s_mag + pos_local + 19 + t / 5 + q;
        int corr = 0;
        if (lane < 48) corr = s_coef[sub][0] * s[0] + s_coef[sub][1] * s[1] + s_coef[sub][2] * s[2] + s_coef[sub][3] * s[3];
        lo = __brevll(__ballot(corr > 0 && lane < 48)) >> 16;
    }
    const uint32_t aa = (uint32_t) (hi >> 32) & 0xffffffu;   // getbits(msg, 9, 32)

    int sk = -2, su = -2, fb0 = 0xff, fb1 = 0xff;
    uint32_t addr = 0, flags = is_long ? REC_LONG : 0;
    bool emit = false;

    if (is_long) {
        // CRC-24 syndrome of the 112-bit frame: lane j < 24 owns syndrome bit j
        const int par = __popcll(hi & lc.PH) + __popcll(lo & lc.PL);
        const uint32_t synd = (uint32_t) __ballot(par & 1) & 0xffffffu;
        bool handled = false;
        // fixDF17msgtype (mode_s.c:276-301): DF one bit away from 17 and the frame is clean once DF := 17
        if (p.fix_df && (df == 1 || df == 16 || df == 19 || df == 21 || df == 25)) {
            const int j = 4 - (__ffs(df ^ 17u) - 1);   // frame bit that differs
            if (synd == p.bit_syndrome[j]) {
                sk = 1800 / 2; su = 1400 / 2; addr = aa;
                flags |= REC_ACCEPT_IF_UNKNOWN | REC_DFFIX | (1u << REC_CORR_SHIFT);
                fb0 = j;
                emit = handled = true;
            }
        }
        if (!handled) {
            if ((hi >> 8) == 0) {
                // first 7 bytes all zero -> -2 (mode_s.c:337-338)
            } else if (df == 16 || df == 20 || df == 21) {
                sk = 1000; su = -1; addr = synd; flags |= REC_COND; emit = true;      // Address/Parity
            } else if (df == 17 || df == 18) {
                int b0 = 0xff, b1 = 0xff;
                const int nerr = synd == 0 ? 0 : wave_diagnose(p.tab_long, p.n_long, synd, b0, b1);
                if (nerr >= 0) {
                    uint32_t a2 = aa;
                    if (nerr >= 1) a2 = fix_aa(a2, b0);
                    if (nerr >= 2) a2 = fix_aa(a2, b1);
                    sk = 1800 / (nerr + 1); su = 1400 / (nerr + 1); addr = a2;
                    flags |= (uint32_t) nerr << REC_CORR_SHIFT;
                    if (a2 == aa) flags |= REC_ACCEPT_IF_UNKNOWN;   // mode_s.c:560: only a changed AA needs the filter
                    if (nerr == 0 && df == 17) flags |= REC_ADDER;
                    if (nerr >= 1) fb0 = b0;
                    if (nerr >= 2) fb1 = b1;
                    emit = true;
                }
            }
            // unrepaired DF 1, 19, 25: scoreModesMessage's default case -> -2
        }
    } else {
        if ((hi >> 8) == 0) return 0;
        const int par = __popcll(hi & lc.PS);
        const uint32_t synd = (uint32_t) __ballot(par & 1) & 0xffffffu;
        if (df == 11) {
            if (synd & 0xffff80u) {
                int b0 = 0xff, b1 = 0xff;
                const int nerr = wave_diagnose(p.tab_short, p.n_short, synd, b0, b1);
                if (nerr == 1) {                                  // 2-bit errors are ambiguous in DF11
                    sk = 800; su = -1; addr = fix_aa(aa, b0);
                    flags |= REC_COND | (1u << REC_CORR_SHIFT);
                    fb0 = b0;
                    emit = true;
                }
            } else if ((synd & 0x7f) == 0) {
                sk = 1600; su = 750; addr = aa; flags |= REC_ACCEPT_IF_UNKNOWN | REC_ADDER; emit = true;
            } else {
                sk = 1000; su = -1; addr = aa; flags |= REC_COND; emit = true;
            }
        } else {   // DF 0, 4, 5
            sk = 1000; su = -1; addr = synd; flags |= REC_COND; emit = true;
        }
    }
    if (!emit) return 0;
    if (lane == 0) {
        u32x4 a, b;
        a.x = pos;
        a.y = (uint32_t) t | (flags << 8) | ((uint32_t) (uint16_t) sk << 16);
        a.z = (uint32_t) (uint16_t) su | ((uint32_t) fb0 << 16) | ((uint32_t) fb1 << 24);
        a.w = addr;
        // msg bytes 0..13: byte b = frame bits 8b..8b+7; memory order = byte 0 first
        const uint64_t h = __builtin_bswap64(hi);
        const uint64_t l = __builtin_bswap64(lo << 16);
        b.x = (uint32_t) h;
        b.y = (uint32_t) (h >> 32);
        b.z = (uint32_t) l;
        b.w = (uint32_t) (l >> 32) & 0xffffu;
        u32x4 *d = (u32x4 *) slot;
        d[0] = a;
        d[1] = b;
        if ((flags & REC_ADDER) && !(__hip_atomic_load(&p.adder_bitmap[addr >> 5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & (1u << (addr & 31))))
            atomicOr(&p.adder_bitmap[addr >> 5], 1u << (addr & 31));   // look first: device-scope RMWs serialise per address
    }
    return flags | 0x100u;
}

// Workgroup = one unit of kTilesPerUnit tiles, processed tile after tile:
//   1. stage kTile + kHalo magnitudes in LDS (coalesced 16-byte loads)
//   2. sweep: every thread evaluates 8 consecutive positions from a 26-sample register window
//      (pre-check + the three threshold tests of demod_2400.c:311-378) -> 3-bit phase mask
//   3. compact the candidates, in position order, into an LDS queue (wave prefix sums)
//   4. slice + score: one wave per candidate, records of a 64-candidate batch are gathered in
//      LDS in (position, phase) order and flushed to the global pool as one segment
__global__ __launch_bounds__(kBlock) void k_sweep_slice_v1(SweepParams p) {
    __shared__ __attribute__((aligned(16))) uint16_t s_mag[kTile + kHalo + 8];
    __shared__ uint16_t s_queue[kTile];
    __shared__ __attribute__((aligned(16))) PhaseRec s_slots[kBatch * 5];
    __shared__ uint32_t s_cls[kTile / 32];
    __shared__ int s_segcount[8];
    __shared__ int s_coef[5][4];
    __shared__ unsigned long long s_cnt[CNT_NUM];

    const int tid = threadIdx.x, lane = lane_id(), wv = tid >> 6;
    if (tid < 20) s_coef[tid >> 2][tid & 3] = c_slice_coef[tid >> 2][tid & 3];
    if (tid < CNT_NUM) s_cnt[tid] = 0;
    LaneConst lc;
    lc.PH = lane < 24 ? p.parity[lane] : 0;
    lc.PL = lane < 24 ? p.parity[24 + lane] : 0;
    lc.PS = lane < 24 ? p.parity[48 + lane] : 0;

    uint32_t n_cand = 0, n_ph[3] = {0, 0, 0};     // per-thread sweep counters
    uint32_t n_cond = 0, n_uncond = 0, n_rec = 0;  // wave-uniform (counted by lane 0)

    for (uint32_t unit = blockIdx.x; unit < p.nunits; unit += gridDim.x) {
        uint32_t prev_hdr = kNone, unit_records = 0;   // meaningful in wave 0 only
        if (tid == 0) p.unit_first[unit] = kNone;
        for (int tile = 0; tile < kUnit / kTile; ++tile) {
            const uint64_t D0 = (uint64_t) unit * kUnit + (uint64_t) tile * kTile;
            if (D0 >= p.n) break;
            __syncthreads();   // previous tile fully consumed
            // ---- 1. stage ----
            for (int i = tid; i < (kTile + kHalo) / 8; i += kBlock)
                *(u32x4 *) &s_mag[8 * i] = *(const u32x4 *) &p.mag[D0 + 8 * i];
            if (tid < kTile / 32) s_cls[tid] = 0;
            __syncthreads();
            // ---- 2. sweep ----
            uint32_t fl[2];
            int pre[2];
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int p0 = it * (kTile / 2) + tid * 8;
                uint32_t w[13];
                {
                    const u32x4 a = *(const u32x4 *) &s_mag[p0], b = *(const u32x4 *) &s_mag[p0 + 8],
                                c = *(const u32x4 *) &s_mag[p0 + 16];
                    w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w;
                    w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
                    w[8] = c.x; w[9] = c.y; w[10] = c.z; w[11] = c.w;
                    w[12] = *(const uint32_t *) &s_mag[p0 + 24];
                }
                uint32_t f = 0;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
#define SM(i) ((int) ((w[((e) + (i)) >> 1] >> ((((e) + (i)) & 1) * 16)) & 0xffffu))
                    const bool pc = SM(1) > SM(7) && SM(12) > SM(14) && SM(12) > SM(15);
                    const int base_noise = SM(5) + SM(8) + SM(16) + SM(17) + SM(18);
                    const int ref = (base_noise * p.thr) >> 5;
                    const int d23 = SM(2) - SM(3), s14 = SM(1) + SM(4), d1011 = SM(10) - SM(11);
                    const int common = s14 - d23 + SM(9) + SM(12);
                    uint32_t m = 0;
                    if (common - d1011 >= ref) m |= 1;
                    if (common + d1011 >= ref) m |= 2;
                    if (s14 + 2 * d23 + d1011 + SM(12) >= ref) m |= 4;
#undef SM
                    if (!pc || D0 + p0 + e >= p.n) m = 0;
                    f |= m << (3 * e);
                }
                fl[it] = f;
                const uint32_t nz = (f | (f >> 1) | (f >> 2)) & 0x249249u;
                const int cnt = __popc(nz);
                n_cand += cnt;
                n_ph[0] += __popc(f & 0x249249u);
                n_ph[1] += __popc((f >> 1) & 0x249249u);
                n_ph[2] += __popc((f >> 2) & 0x249249u);
                int total;
                pre[it] = wave_excl_scan(cnt, total);
                if (lane == 0) s_segcount[it * 4 + wv] = total;
            }
            __syncthreads();
            // ---- 3. ordered compaction ----
            int ncand = 0;
            {
                int segbase[8];
#pragma unroll
                for (int sgi = 0; sgi < 8; ++sgi) { segbase[sgi] = ncand; ncand += s_segcount[sgi]; }
#pragma unroll
                for (int it = 0; it < 2; ++it) {
                    int dst = segbase[it * 4 + wv] + pre[it];
                    const int p0 = it * (kTile / 2) + tid * 8;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const uint32_t m = (fl[it] >> (3 * e)) & 7u;
                        if (m) s_queue[dst++] = (uint16_t) (((p0 + e) << 3) | m);
                    }
                }
            }
            __syncthreads();
            // ---- 4. slice + score, 64 candidates per batch ----
            for (int b0 = 0; b0 < ncand; b0 += kBatch) {
                for (int i = tid; i < kBatch * 5; i += kBlock) s_slots[i].phase = 0;
                __syncthreads();
                for (int j = 0; j < kBatch / 4; ++j) {
                    const int ci = b0 + wv * (kBatch / 4) + j;
                    if (ci >= ncand) break;
                    const uint32_t q = s_queue[ci];
                    const int pos_local = q >> 3;
                    const uint32_t mask = q & 7u;
                    const uint32_t pos = (uint32_t) (D0 + pos_local);
                    PhaseRec *slots = &s_slots[(wv * (kBatch / 4) + j) * 5];
                    uint32_t any_uncond = 0, any_cond = 0, nrec = 0;
#pragma unroll
                    for (int t = 4; t <= 8; ++t) {
                        const uint32_t need = t <= 5 ? (mask & 1u) : t <= 7 ? (mask & 2u) : (mask & 4u);
                        if (!need) continue;
                        const uint32_t r = slice_and_score(p, lc, s_mag, s_coef, pos_local, pos, t, &slots[t - 4]);
                        if (r) {
                            ++nrec;
                            if (r & REC_COND) any_cond = 1; else any_uncond = 1;
                        }
                    }
                    n_rec += nrec;
                    if (any_uncond) ++n_uncond;
                    else if (any_cond) {
                        ++n_cond;
                        if (lane == 0) atomicOr(&s_cls[pos_local >> 5], 1u << (pos_local & 31));
                    }
                }
                __syncthreads();
                if (wv == 0) {
                    // flush the batch's records, in slot order, as one segment of the unit's chain
                    uint64_t valid[5];
                    int cnt = 0;
#pragma unroll
                    for (int r = 0; r < 5; ++r) {
                        valid[r] = __ballot(s_slots[r * WAVE + lane].phase != 0);
                        cnt += __popcll(valid[r]);
                    }
                    if (cnt > 0) {
                        uint32_t base = 0;
                        if (lane == 0) base = atomicAdd(p.pool_used, (uint32_t) cnt + 1u);
                        base = rfl(base);
                        if ((uint64_t) base + cnt + 1 > p.pool_cap) {
                            if (lane == 0) atomicAdd(&p.counters[CNT_POOL_OVERFLOW], 1ull);
                        } else {
                            if (lane == 0) {
                                u32x4 h0 = {(uint32_t) cnt, 0xFFu, 0u, kNone}, h1 = {0, 0, 0, 0};
                                u32x4 *hd = (u32x4 *) &p.pool[base];
                                hd[0] = h0; hd[1] = h1;
                                if (prev_hdr == kNone) p.unit_first[unit] = base;
                                else p.pool[prev_hdr].addr = base;
                            }
                            prev_hdr = base;
                            uint32_t run = base + 1;
#pragma unroll
                            for (int r = 0; r < 5; ++r) {
                                if ((valid[r] >> lane) & 1) {
                                    const uint32_t dst = run + __popcll(valid[r] & ((1ull << lane) - 1));
                                    const u32x4 *src = (const u32x4 *) &s_slots[r * WAVE + lane];
                                    u32x4 *d = (u32x4 *) &p.pool[dst];
                                    d[0] = src[0]; d[1] = src[1];
                                }
                                run += __popcll(valid[r]);
                            }
                            unit_records += cnt;
                        }
                    }
                }
                __syncthreads();
            }
            // ---- class bitmap of this tile (all words, so the bitmap needs no clearing) ----
            if (tid < kTile / 32) p.class_bitmap[(D0 >> 5) + tid] = s_cls[tid];
        }
        if (tid == 0) p.unit_count[unit] = unit_records;
    }
    // ---- counters: one set of atomics per workgroup ----
    atomicAdd(&s_cnt[CNT_CANDIDATES], (unsigned long long) n_cand);
    atomicAdd(&s_cnt[CNT_PHASE0 + 0], (unsigned long long) n_ph[0]);
    atomicAdd(&s_cnt[CNT_PHASE0 + 2], (unsigned long long) n_ph[1]);
    atomicAdd(&s_cnt[CNT_PHASE0 + 4], (unsigned long long) n_ph[2]);
    if (lane == 0) {
        atomicAdd(&s_cnt[CNT_RECORDS], (unsigned long long) n_rec);
        atomicAdd(&s_cnt[CNT_CLASS_COND], (unsigned long long) n_cond);
        atomicAdd(&s_cnt[CNT_CLASS_UNCOND], (unsigned long long) n_uncond);
    }
    __syncthreads();
    if (tid < CNT_NUM && s_cnt[tid]) {
        unsigned long long v = s_cnt[tid];
        if (tid == CNT_PHASE0 + 0) { atomicAdd(&p.counters[CNT_PHASE0 + 0], v); atomicAdd(&p.counters[CNT_PHASE0 + 1], v); }
        else if (tid == CNT_PHASE0 + 2) { atomicAdd(&p.counters[CNT_PHASE0 + 2], v); atomicAdd(&p.counters[CNT_PHASE0 + 3], v); }
        else atomicAdd(&p.counters[tid], v);
    }
}

void launch_sweep_slice_v1(const SweepParams &p, hipStream_t s) {
    if (p.nunits == 0) return;
    unsigned blocks = p.nunits < 256u * 5u ? p.nunits : 256u * 5u;
    hipLaunchKernelGGL(k_sweep_slice_v1, dim3(blocks), dim3(kBlock), 0, s, p);
}

// ---------------------------------------------------------------------------------------------
// k_sweep_slice (second generation): same sweep, but the slicer runs one LANE per
// (candidate, phase) pair instead of one wave per candidate, so 64 frames are sliced at once
// and nothing waits on ballots or scalar branches:
//
//   sweep step (2048 positions)  -> ordered candidate queue (u16: pos_local<<3 | phase mask)
//   drain:  expand 256 candidates into their (position, phase) pairs (block prefix sum)
//           stage A  lane = pair: slice frame bits 0..4 = the DF field; keep pairs whose DF is in
//                    valid_df_{long,short}_bitset (demod_2400.c:223-238); ordered append to a ring
//           stage B  lane = surviving pair, 256 at a time (always full waves): slice the rest of
//                    the frame five bits at a time, row-major over the five correlators so the
//                    coefficients are immediates (the phase only changes per-lane sample offsets),
//                    CRC-24 syndrome by XOR of per-group table entries (GF(2)-linear), score;
//                    ordered append of the records to an LDS staging area
//           flush    one atomicAdd per segment, coalesced 32-byte record copies
//
// Frame bit k of try-phase t uses correlator row (t + 2k) % 5 at sample pa + 19 + t/5 +
// (t%5 + 12k)/5 (closed form of slice_byte, demod_2400.c:133-213).  With k = 5g + i the row is
// (p + 2i) % 5, p = t % 5, so inside every group of five bits each row is used exactly once:
// iterating rows 0..4 (compile-time coefficients) visits bit i_r = 3(r - p) mod 5 of the group at
// sample offset 12g + (p + 12 i_r)/5.
// ---------------------------------------------------------------------------------------------
constexpr int kSub = kBlock * 8;            // positions per sweep step
constexpr int kCQCap = kSub + 512;          // candidate queue: a whole worst-case step fits after a drain
constexpr int kPairCap = kBlock * 5;        // pairs of one 256-candidate expansion
constexpr int kVCap = 512;                  // ring of valid-DF pairs waiting for stage B (power of two)
// Publish "a clean DF17 / DF11-IID0 frame of this stream carries `addr`" (mode_s.c:766-779) in the
// 2^24-bit adder bitmap.  Device-scope RMWs execute at the memory side (the 8 XCD L2s are not
// coherent with each other) and serialise per address; a few hundred aircraft addresses are hit
// millions of times, and sending every sighting doubled the sweep kernel's run time.  So: a
// direct-mapped LDS cache of what this workgroup already published, and — bits are only ever
// set — a look (agent-scope load) before the RMW.
template <int CACHE>
__device__ __forceinline__ void adder_publish(uint32_t *bitmap, uint32_t *cache, uint32_t addr) {
    const uint32_t h = (addr ^ (addr >> 10) ^ (addr >> 17)) & (CACHE - 1);
    if (cache[h] == addr) return;
    cache[h] = addr;
    const uint32_t bit = 1u << (addr & 31);
    if (!(__hip_atomic_load(&bitmap[addr >> 5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & bit)) atomicOr(&bitmap[addr >> 5], bit);
}

constexpr int kAdderCache = 512;            // direct-mapped LDS cache of adder addresses already published
constexpr int kPoolChunk = 1024;            // v2: pool records a workgroup reserves per returning atomic
constexpr int kQuadsPerRound = kBlock / 4;  // frames sliced per stage-B round (4 lanes each)
constexpr int kStageCap = 192;              // staged records; flushed when more than 128 are waiting

struct SliceGeom {          // per lane, derived from the try-phase
    int wi[5];              // dword index (into the LDS tile viewed as u32) of the even sample at or below row r's first tap
    int par[5];             // 0 / 16: first tap is the low / high half of that dword
    int sh[5];              // position (4 - i_r) of row r's bit inside the 5-bit group value
};

__device__ __forceinline__ void make_geom(int pos_local, int t, SliceGeom &g) {
    const int p = t % 5;
    const int base = pos_local + 19 + t / 5;
#pragma unroll
    for (int r = 0; r < 5; ++r) {
        int i = (r - p + 5) * 3;
        i -= (i / 5) * 5;
        const int s = base + (p + 12 * i) / 5;
        g.wi[r] = s >> 1;
        g.par[r] = (s & 1) * 16;
        g.sh[r] = 4 - i;
    }
}

// Three (four for row 4) consecutive u16 taps starting at a per-lane sample index of either parity,
// fetched as ALIGNED dwords and funnel-shifted: hipcc otherwise fuses adjacent u16 LDS loads into
// 2-byte-aligned ds_read_b32, which the LDS replays lane by lane.
#define TAPS3(R, GW)                                                                   \
    const uint32_t a_ = w32[g.wi[R] + (GW)], b_ = w32[g.wi[R] + (GW) + 1];             \
    const uint32_t x_ = __builtin_amdgcn_alignbit(b_, a_, g.par[R]);                   \
    const int m0 = (int) (x_ & 0xffffu), m1 = (int) (x_ >> 16), m2 = (int) ((b_ >> g.par[R]) & 0xffffu);

// the five correlators (slice_phase0..4, demod_2400.c:74-93); gw = 6 * group (dwords per 12 samples)
__device__ __forceinline__ uint32_t slice_group(const uint32_t *w32, const SliceGeom &g, int gw) {
    uint32_t v = 0;
    { TAPS3(0, gw) v |= (uint32_t) (18 * m0 - 15 * m1 - 3 * m2 > 0) << g.sh[0]; }
    { TAPS3(1, gw) v |= (uint32_t) (14 * m0 - 5 * m1 - 9 * m2 > 0) << g.sh[1]; }
    { TAPS3(2, gw) v |= (uint32_t) (16 * m0 + 5 * m1 - 20 * m2 > 0) << g.sh[2]; }
    { TAPS3(3, gw) v |= (uint32_t) (7 * m0 + 11 * m1 - 18 * m2 > 0) << g.sh[3]; }
    {
        TAPS3(4, gw)
        const uint32_t c_ = w32[g.wi[4] + gw + 2];
        const int m3 = (int) ((__builtin_amdgcn_alignbit(c_, b_, g.par[4]) >> 16) & 0xffffu);
        v |= (uint32_t) (4 * m0 + 15 * m1 - 20 * m2 + m3 > 0) << g.sh[4];
    }
    return v;
}

// The same five correlators on a tile whose samples are stored with the top bit flipped (u16 m -> i16 m - 32768),
// so that two taps go through one v_dot2_i32_i16.  sum(c_j * m_j) > 0  <=>  sum(c_j * (m_j - 32768)) > -32768 * sum(c_j),
// and the rows' coefficient sums are 0, 0, 1, 0, 0 (demod_2400.c:74-93).
typedef short v2i16 __attribute__((ext_vector_type(2)));
// All eleven LDS dwords of the group are requested before the first is used: the five bits are independent, and
// one round trip to the LDS per group instead of five is what the slicer's speed hangs on.
__device__ __forceinline__ uint32_t slice_group_biased(const uint32_t *w32, const SliceGeom &g, int gw) {
    uint32_t a[5], b[5];
#pragma unroll
    for (int r = 0; r < 5; ++r) { a[r] = w32[g.wi[r] + gw]; b[r] = w32[g.wi[r] + gw + 1]; }
    const uint32_t c4 = w32[g.wi[4] + gw + 2];
    __builtin_amdgcn_sched_barrier(0);          // keep the loads above, the arithmetic below
    // NEGATED coefficients: the bit is then the sign bit of the sum — one shift and one v_lshl_or per bit instead of
    // compare, select, shift, or.  Row 2's threshold (sum > -32768) goes into its accumulator: -(sum) - 32768 < 0.
    const v2i16 k0a = {-18, 15}, k0b = {3, 0}, k1a = {-14, 5}, k1b = {9, 0}, k2a = {-16, -5}, k2b = {20, 0},
                k3a = {-7, -11}, k3b = {18, 0}, k4a = {-4, -15}, k4b = {20, -1};
#define PK01(R) __builtin_bit_cast(v2i16, __builtin_amdgcn_alignbit(b[R], a[R], g.par[R]))   /* taps 0, 1 */
#define PK2(R) __builtin_bit_cast(v2i16, b[R] >> g.par[R])                                    /* tap 2 in the low half */
    const int s0 = __builtin_amdgcn_sdot2(PK2(0), k0b, __builtin_amdgcn_sdot2(PK01(0), k0a, 0, false), false);
    const int s1 = __builtin_amdgcn_sdot2(PK2(1), k1b, __builtin_amdgcn_sdot2(PK01(1), k1a, 0, false), false);
    const int s2 = __builtin_amdgcn_sdot2(PK2(2), k2b, __builtin_amdgcn_sdot2(PK01(2), k2a, -32768, false), false);
    const int s3 = __builtin_amdgcn_sdot2(PK2(3), k3b, __builtin_amdgcn_sdot2(PK01(3), k3a, 0, false), false);
    const v2i16 y4 = __builtin_bit_cast(v2i16, __builtin_amdgcn_alignbit(c4, b[4], g.par[4]));   // taps 2, 3
    const int s4 = __builtin_amdgcn_sdot2(y4, k4b, __builtin_amdgcn_sdot2(PK01(4), k4a, 0, false), false);
#undef PK01
#undef PK2
    uint32_t v = ((uint32_t) s0 >> 31) << g.sh[0];
    v |= ((uint32_t) s1 >> 31) << g.sh[1];
    v |= ((uint32_t) s2 >> 31) << g.sh[2];
    v |= ((uint32_t) s3 >> 31) << g.sh[3];
    v |= ((uint32_t) s4 >> 31) << g.sh[4];
    return v;
}
#undef TAPS3

// per-lane modesChecksumDiagnose (crc.c:383-406): binary search over the sorted syndromes, which
// are staged in LDS (a miss — the common case for noise — never touches global memory; dependent
// global loads under a streaming kernel cost thousands of cycles each); a hit fetches the packed
// entry (syndrome<<16 | bit0<<8 | bit1) from the global table.
__device__ __forceinline__ int lane_diagnose(const uint32_t *keys, const uint64_t *tab, int n, uint32_t synd, int &b0, int &b1) {
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (keys[mid] < synd) lo = mid + 1; else hi = mid;
    }
    if (lo >= n || keys[lo] != synd) return -1;
    const uint64_t e = tab[lo];
    b0 = (int) ((e >> 8) & 0xff);
    b1 = (int) (e & 0xff);
    return b1 == 0xff ? 1 : 2;
}

// Frame bits 5..114 = groups 1..22 of five bits.  FOUR adjacent lanes share one frame: lane j of
// the quad slices groups 1+6j .. 6+6j (j = 3: groups 19..22) into a 30-bit chunk and a partial
// syndrome (CRC-24 is GF(2)-linear: the syndrome is the XOR of per-group table entries), so a
// 256-thread round handles 64 frames with every wave busy and a 6-iteration dependent chain per
// lane instead of 22.  The loop stays rolled: the unrolled form is ~11 KB of straight-line code
// per inlined copy.  Lanes of a 56-bit frame stop after group 11 and contribute zeros.
template <bool BIASED = false>
__device__ __forceinline__ void slice_chunk(const uint32_t *w32, const uint32_t *s_gsyn, const SliceGeom &g, bool is_long,
                                            int j, uint32_t &chunk, uint32_t &synd) {
    const uint32_t *gs = s_gsyn + (is_long ? 0 : kGroupsLong * 32);
    const int g0 = 1 + 6 * j;
    const int ng = j == 3 ? 4 : 6;
    chunk = 0;
#pragma unroll 2
    for (int k = 0; k < ng; ++k) {
        const int G = g0 + k;
        uint32_t grp = 0;
        if (is_long || G <= 11) {
            grp = BIASED ? slice_group_biased(w32, g, 6 * G) : slice_group(w32, g, 6 * G);
            if (G == 22) grp &= 0x18u;                    // frame bits 110, 111 only
            if (G == 11 && !is_long) grp &= 0x10u;        // frame bit 55 only
            synd ^= gs[G * 32 + grp];
        }
        chunk = (chunk << 5) | grp;
    }
}

// value of `v` in lane k of the caller's quad (DPP quad_perm broadcast, no LDS traffic)
template <int K>
__device__ __forceinline__ uint32_t quad_bcast(uint32_t v) {
    return (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, K * 0x55, 0xf, 0xf, false);
}

struct BlockScan {            // ordered offsets across the 4 waves of the workgroup
    int *wcount;              // LDS [4]
    // flag version (1 bit per thread); one __syncthreads inside
    __device__ __forceinline__ int flags(bool f, int &total) {
        const uint64_t m = __ballot(f);
        const int lane = lane_id(), wv = threadIdx.x >> 6;
        if (lane == 0) wcount[wv] = __popcll(m);
        __syncthreads();
        int off = 0, tot = 0;
#pragma unroll
        for (int k = 0; k < kBlock / WAVE; ++k) { const int c = wcount[k]; if (k < wv) off += c; tot += c; }
        total = tot;
        return off + __popcll(m & ((1ull << lane) - 1));
    }
    __device__ __forceinline__ int counts(int v, int &total) {
        int wt;
        const int ex = wave_excl_scan(v, wt);
        const int lane = lane_id(), wv = threadIdx.x >> 6;
        if (lane == 0) wcount[wv] = wt;
        __syncthreads();
        int off = 0, tot = 0;
#pragma unroll
        for (int k = 0; k < kBlock / WAVE; ++k) { const int c = wcount[k]; if (k < wv) off += c; tot += c; }
        total = tot;
        return off + ex;
    }
};

__global__ __launch_bounds__(kBlock) void k_sweep_slice_v2(SweepParams p) {
    __shared__ __attribute__((aligned(16))) uint16_t s_mag[kTile2 + kHalo + 8];
    __shared__ __attribute__((aligned(16))) PhaseRec s_stage[kStageCap];
    __shared__ uint16_t s_cq[kCQCap];
    __shared__ uint16_t s_pairs[kPairCap];
    __shared__ uint32_t s_v[kVCap];
    __shared__ uint32_t s_gsyn[(kGroupsLong + kGroupsShort) * 32];
    __shared__ uint32_t s_cls_cond[kTile2 / 32], s_cls_uncond[kTile2 / 32];
    __shared__ int s_wcount[2][kBlock / WAVE];
    __shared__ uint32_t s_bcast;
    __shared__ unsigned long long s_cnt[CNT_NUM];
    __shared__ uint32_t s_acache[kAdderCache];
    extern __shared__ uint32_t s_keys[];   // n_long + n_short sorted syndromes (dynamic: 0.6 KB for --fix, 20 KB for --aggressive)

    const int tid = threadIdx.x;
    for (int i = tid; i < (kGroupsLong + kGroupsShort) * 32; i += kBlock) s_gsyn[i] = p.group_syndrome[i];
    for (int i = tid; i < kAdderCache; i += kBlock) s_acache[i] = 0xFFFFFFFFu;
    for (int i = tid; i < p.n_long; i += kBlock) s_keys[i] = (uint32_t) (p.tab_long[i] >> 16);
    for (int i = tid; i < p.n_short; i += kBlock) s_keys[p.n_long + i] = (uint32_t) (p.tab_short[i] >> 16);
    if (tid < CNT_NUM) s_cnt[tid] = 0;
    uint32_t n_cand = 0, n_ph[3] = {0, 0, 0}, n_cls_cond = 0, n_cls_uncond = 0, n_rec = 0;
    uint32_t chunk_base = 0, chunk_left = 0;           // last thread only: reserved pool space
    constexpr int kPre = ((kTile2 + kHalo) / 8 + (kBlock - WAVE) - 1) / (kBlock - WAVE);   // 16-byte loads per loading thread and tile
    u32x4 pre[kPre];
    bool have_pre = false;
    long long dbg_b = 0, dbg_nb = 0, dbg_take = 0, dbg_s1 = 0, dbg_s2 = 0, dbg_s3 = 0;
    long long dbg_tile[6] = {0, 0, 0, 0, 0, 0};
    const long long dbg_t0 = DBG_CLOCK();
    int scan_sel = 0;   // alternate between two wave-count arrays so back-to-back scans need no extra barrier
    auto scan = [&]() __attribute__((always_inline)) { BlockScan b{s_wcount[scan_sel]}; scan_sel ^= 1; return b; };

    for (uint32_t unit = blockIdx.x; unit < p.nunits; unit += gridDim.x) {
        uint32_t prev_hdr = kNone, unit_records = 0;   // last thread only
        int scount = 0;                                 // staged records (uniform)
        if (tid == kBlock - 1) p.unit_first[unit] = kNone;

        // Flush the staged records as one segment of the unit's chain.  Only the LAST wave issues
        // global stores (records, headers, class bitmap): vector-memory returns are in order on gfx9,
        // so a wave that has stores in flight waits for their acknowledgements before it sees the
        // next tile's load data; waves 0..2 do all the tile loads and never store.
        // Pool space is reserved kPoolChunk records at a time (one returning atomic per chunk, not
        // per segment: a single hot word is a memory-side atomic every workgroup would queue on).
        auto flush = [&]() __attribute__((always_inline)) {
            if (scount == 0) return;
            if (tid == kBlock - 1) {
                if (chunk_left < (uint32_t) scount + 1u) {
                    chunk_base = atomicAdd(p.pool_used, (uint32_t) kPoolChunk);
                    chunk_left = kPoolChunk;
                }
                s_bcast = chunk_base;
                chunk_base += (uint32_t) scount + 1u;
                chunk_left -= (uint32_t) scount + 1u;
            }
            __syncthreads();
            const uint32_t base = s_bcast;
            const bool ok = (uint64_t) base + kPoolChunk <= p.pool_cap;
            if (tid >= kBlock - WAVE) {
                if (ok) {
                    for (int i = tid - (kBlock - WAVE); i < scount; i += WAVE) {
                        const u32x4 *src = (const u32x4 *) &s_stage[i];
                        u32x4 *d = (u32x4 *) &p.pool[base + 1 + i];
                        d[0] = src[0]; d[1] = src[1];
                    }
                    if (tid == kBlock - 1) {
                        u32x4 h0 = {(uint32_t) scount, 0xFFu, 0u, kNone}, h1 = {0, 0, 0, 0};
                        u32x4 *hd = (u32x4 *) &p.pool[base];
                        hd[0] = h0; hd[1] = h1;
                        if (prev_hdr == kNone) p.unit_first[unit] = base; else p.pool[prev_hdr].addr = base;
                        prev_hdr = base;
                        unit_records += scount;
                        n_rec += scount;
                    }
                } else if (tid == kBlock - 1) {
                    atomicAdd(&p.counters[CNT_POOL_OVERFLOW], 1ull);
                }
            }
            scount = 0;
            __syncthreads();
        };

        for (int tile = 0; tile < kUnit / kTile2; ++tile) {
            const uint64_t D0 = (uint64_t) unit * kUnit + (uint64_t) tile * kTile2;
            if (D0 >= p.n) break;
            const long long tt0 = DBG_CLOCK();
            __syncthreads();
            const long long tta = DBG_CLOCK();
            // the tile was prefetched into registers while the previous tile was being sliced
            if (tid < kBlock - WAVE) {
                if (!have_pre) {
#pragma unroll
                    for (int k = 0; k < kPre; ++k) {
                        const int i = tid + k * (kBlock - WAVE);
                        if (i < (kTile2 + kHalo) / 8) pre[k] = *(const u32x4 *) &p.mag[D0 + 8 * i];
                    }
                }
#pragma unroll
                for (int k = 0; k < kPre; ++k) {
                    const int i = tid + k * (kBlock - WAVE);
                    if (i < (kTile2 + kHalo) / 8) *(u32x4 *) &s_mag[8 * i] = pre[k];
                }
            }
            if (tid < kTile2 / 32) { s_cls_cond[tid] = 0; s_cls_uncond[tid] = 0; }
            {   // prefetch the workgroup's next tile (same unit, or the first tile of its next unit)
                uint64_t Dn = D0 + kTile2;
                if (tile + 1 >= kUnit / kTile2 || Dn >= p.n) Dn = (uint64_t) (unit + gridDim.x) * kUnit;
                have_pre = (uint64_t) (unit + gridDim.x) * kUnit == Dn ? (unit + gridDim.x < p.nunits) : true;
                if (have_pre && tid < kBlock - WAVE) {
#pragma unroll
                    for (int k = 0; k < kPre; ++k) {
                        const int i = tid + k * (kBlock - WAVE);
                        if (i < (kTile2 + kHalo) / 8) pre[k] = *(const u32x4 *) &p.mag[Dn + 8 * i];
                    }
                }
            }
#if MGPU_KERNEL_TIMERS
            __builtin_amdgcn_s_waitcnt(0);
#endif
            const long long ttb = DBG_CLOCK();
            __syncthreads();
            const long long tt1 = DBG_CLOCK();
            if (tid == 0) { dbg_tile[5] += tta - tt0; dbg_s3 += ttb - tta; }

            int ccount = 0;            // queued candidates (uniform)
            int vhead = 0, vcount = 0; // ring of valid pairs (uniform)

            // ---- stage B over the first `take` ring entries ----
            auto stage_b = [&](int take) __attribute__((always_inline)) {
                const long long tb0 = DBG_CLOCK();
                long long tb1 = tb0;
                bool emit = false;
                u32x4 ra = {0, 0, 0, 0}, rb = {0, 0, 0, 0};
                const int quad = tid >> 2, qj = tid & 3;
                const bool have = quad < take;
                uint32_t e = 0, chunk = 0, psyn = 0, df = 0;
                bool is_long = false;
                int pos_local = 0, t = 4;
                if (have) {
                    e = s_v[(vhead + quad) & (kVCap - 1)];
                    pos_local = (int) ((e & 0xffffu) >> 3);
                    t = 4 + (int) (e & 7u);
                    df = e >> 16;
                    is_long = (p.valid_long >> df) & 1;
                    if (!(p.debug_stage & 16) && (is_long || qj < 2)) {
                        SliceGeom g;
                        make_geom(pos_local, t, g);
                        slice_chunk((const uint32_t *) s_mag, s_gsyn, g, is_long, qj, chunk, psyn);
                    }
                }
                // gather the quad's chunks / syndrome in every lane (lane 0 of the quad goes on alone)
                const uint32_t c0 = quad_bcast<0>(chunk), c1 = quad_bcast<1>(chunk), c2 = quad_bcast<2>(chunk), c3 = quad_bcast<3>(chunk);
                const uint32_t sx = quad_bcast<0>(psyn) ^ quad_bcast<1>(psyn) ^ quad_bcast<2>(psyn) ^ quad_bcast<3>(psyn);
                if (have && qj == 0) {
                    uint32_t W[4];
                    W[0] = (df << 27) | (c0 >> 3);
                    W[1] = ((c0 & 7u) << 29) | (c1 >> 1);
                    W[2] = ((c1 & 1u) << 31) | (c2 << 1) | (c3 >> 19);
                    W[3] = (c3 << 13) & 0xffff0000u;
                    const uint32_t synd = sx ^ s_gsyn[(is_long ? 0 : kGroupsLong * 32) + df];
                    tb1 = DBG_CLOCK();
                    const uint32_t aa = W[0] & 0xffffffu;          // getbits(msg, 9, 32)
                    int sk = -2, su = -2, fb0 = 0xff, fb1 = 0xff;
                    uint32_t addr = 0, flags = is_long ? REC_LONG : 0;
                    if (is_long) {
                        bool handled = false;
                        if (p.fix_df && (df == 1 || df == 16 || df == 19 || df == 21 || df == 25)) {
                            const int j = 4 - (__ffs(df ^ 17u) - 1);
                            if (synd == s_gsyn[16u >> j]) {             // == bit_syndrome[j]; fixDF17msgtype, mode_s.c:276-301
                                sk = 900; su = 700; addr = aa;
                                flags |= REC_ACCEPT_IF_UNKNOWN | REC_DFFIX | (1u << REC_CORR_SHIFT);
                                fb0 = j; emit = handled = true;
                            }
                        }
                        if (!handled && !(W[0] == 0 && (W[1] >> 8) == 0)) {
                            if (df == 16 || df == 20 || df == 21) {
                                sk = 1000; su = -1; addr = synd; flags |= REC_COND; emit = true;
                            } else if (df == 17 || df == 18) {
                                int b0 = 0xff, b1 = 0xff;
                                const int nerr = synd == 0 ? 0 : (p.debug_stage & 4) ? -1 : lane_diagnose(s_keys, p.tab_long, p.n_long, synd, b0, b1);
                                if (nerr >= 0) {
                                    uint32_t a2 = aa;
                                    if (nerr >= 1) a2 = fix_aa(a2, b0);
                                    if (nerr >= 2) a2 = fix_aa(a2, b1);
                                    sk = 1800 / (nerr + 1); su = 1400 / (nerr + 1); addr = a2;
                                    flags |= (uint32_t) nerr << REC_CORR_SHIFT;
                                    if (a2 == aa) flags |= REC_ACCEPT_IF_UNKNOWN;
                                    if (nerr == 0 && df == 17) flags |= REC_ADDER;
                                    if (nerr >= 1) fb0 = b0;
                                    if (nerr >= 2) fb1 = b1;
                                    emit = true;
                                }
                            }
                        }
                    } else if (!(W[0] == 0 && (W[1] >> 8) == 0)) {
                        if (df == 11) {
                            if (synd & 0xffff80u) {
                                int b0 = 0xff, b1 = 0xff;
                                if (!(p.debug_stage & 4) && lane_diagnose(s_keys + p.n_long, p.tab_short, p.n_short, synd, b0, b1) == 1) {
                                    sk = 800; su = -1; addr = fix_aa(aa, b0);
                                    flags |= REC_COND | (1u << REC_CORR_SHIFT);
                                    fb0 = b0; emit = true;
                                }
                            } else if ((synd & 0x7f) == 0) {
                                sk = 1600; su = 750; addr = aa; flags |= REC_ACCEPT_IF_UNKNOWN | REC_ADDER; emit = true;
                            } else {
                                sk = 1000; su = -1; addr = aa; flags |= REC_COND; emit = true;
                            }
                        } else {
                            sk = 1000; su = -1; addr = synd; flags |= REC_COND; emit = true;
                        }
                    }
                    if (emit) {
                        ra.x = (uint32_t) (D0 + pos_local);
                        ra.y = (uint32_t) t | (flags << 8) | ((uint32_t) (uint16_t) sk << 16);
                        ra.z = (uint32_t) (uint16_t) su | ((uint32_t) fb0 << 16) | ((uint32_t) fb1 << 24);
                        ra.w = addr;
                        if (!is_long) { W[1] &= 0xffffff00u; W[2] = 0; W[3] = 0; }
                        rb.x = __builtin_bswap32(W[0]); rb.y = __builtin_bswap32(W[1]);
                        rb.z = __builtin_bswap32(W[2]); rb.w = __builtin_bswap32(W[3]) & 0xffffu;
                        if (flags & REC_COND) atomicOr(&s_cls_cond[pos_local >> 5], 1u << (pos_local & 31));
                        else atomicOr(&s_cls_uncond[pos_local >> 5], 1u << (pos_local & 31));
                        if (flags & REC_ADDER) {
                            // Device-scope atomics are executed at the memory side (the 8 XCD L2s are not coherent) and a
                            // few hundred aircraft addresses are hit millions of times: remember what this workgroup
                            // already published in a small LDS cache and only send first sightings.
                            adder_publish<kAdderCache>(p.adder_bitmap, s_acache, addr);
                        }
                    }
                }
                const long long tb2 = DBG_CLOCK();
                int total;
                const int idx = scan().flags(emit, total);
                const long long tb3 = DBG_CLOCK(); (void) tb3;
                if (emit) {
                    u32x4 *d = (u32x4 *) &s_stage[scount + idx];
                    d[0] = ra; d[1] = rb;
                }
                __syncthreads();
                scount += total;
                vhead += take;
                vcount -= take;
                if (p.debug_stage & 8) scount = 0;
                if (scount > kStageCap - kQuadsPerRound) flush();
                if (tid == 0) { dbg_b += DBG_CLOCK() - tb0; dbg_nb += 1; dbg_take += take; dbg_s1 += tb1 - tb0; dbg_s2 += tb2 - tb1; }
            };

            // ---- expand + stage A over all queued candidates, feeding stage B ----
            auto drain = [&]() __attribute__((always_inline)) {
                if (p.debug_stage == 1) { ccount = 0; return; }
                for (int c0 = 0; c0 < ccount; c0 += kBlock) {
                    const int ci = c0 + tid;
                    const uint32_t code = ci < ccount ? s_cq[ci] : 0u;
                    const uint32_t mask = code & 7u;
                    const int np = 2 * (int) (mask & 1u) + (int) (mask & 2u) + (int) ((mask >> 2) & 1u);
                    int npairs;
                    int off = scan().counts(np, npairs);
                    const uint32_t pl = code & 0xfff8u;
                    if (mask & 1u) { s_pairs[off++] = (uint16_t) (pl | 0u); s_pairs[off++] = (uint16_t) (pl | 1u); }
                    if (mask & 2u) { s_pairs[off++] = (uint16_t) (pl | 2u); s_pairs[off++] = (uint16_t) (pl | 3u); }
                    if (mask & 4u) { s_pairs[off++] = (uint16_t) (pl | 4u); }
                    __syncthreads();
                    for (int a0 = 0; a0 < npairs; a0 += kBlock) {
                        const int j = a0 + tid;
                        bool valid = false;
                        uint32_t entry = 0;
                        if (j < npairs) {
                            const uint32_t pc = s_pairs[j];
                            SliceGeom g;
                            make_geom((int) (pc >> 3), 4 + (int) (pc & 7u), g);
                            const uint32_t df = slice_group((const uint32_t *) s_mag, g, 0);
                            valid = ((p.valid_long | p.valid_short) >> df) & 1;
                            entry = pc | (df << 16);
                        }
                        int nvalid;
                        const int idx = scan().flags(valid, nvalid);
                        if (valid) s_v[(vhead + vcount + idx) & (kVCap - 1)] = entry;
                        __syncthreads();
                        vcount += nvalid;
                        if (p.debug_stage == 2) { vcount = 0; vhead = 0; }
                        while (vcount >= kQuadsPerRound) stage_b(kQuadsPerRound);
                    }
                }
                ccount = 0;
            };

            // ---- sweep, kSub positions per step ----
            for (int sub = 0; sub < kTile2 / kSub; ++sub) {
                if (D0 + (uint64_t) sub * kSub >= p.n) break;
                if (ccount + kSub > kCQCap) drain();
                const int p0 = sub * kSub + tid * 8;
                uint32_t w[13];
                {
                    const u32x4 a = *(const u32x4 *) &s_mag[p0], b = *(const u32x4 *) &s_mag[p0 + 8],
                                c = *(const u32x4 *) &s_mag[p0 + 16];
                    w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w;
                    w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
                    w[8] = c.x; w[9] = c.y; w[10] = c.z; w[11] = c.w;
                    w[12] = *(const uint32_t *) &s_mag[p0 + 24];
                }
                uint32_t f = 0;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
#define SM(i) ((int) ((w[((e) + (i)) >> 1] >> ((((e) + (i)) & 1) * 16)) & 0xffffu))
                    const bool pc = SM(1) > SM(7) && SM(12) > SM(14) && SM(12) > SM(15);
                    const int base_noise = SM(5) + SM(8) + SM(16) + SM(17) + SM(18);
                    const int ref = (base_noise * p.thr) >> 5;
                    const int d23 = SM(2) - SM(3), s14 = SM(1) + SM(4), d1011 = SM(10) - SM(11);
                    const int common = s14 - d23 + SM(9) + SM(12);
                    uint32_t m = 0;
                    if (common - d1011 >= ref) m |= 1;
                    if (common + d1011 >= ref) m |= 2;
                    if (s14 + 2 * d23 + d1011 + SM(12) >= ref) m |= 4;
#undef SM
                    if (!pc || D0 + p0 + e >= p.n) m = 0;
                    f |= m << (3 * e);
                }
                const uint32_t nz = (f | (f >> 1) | (f >> 2)) & 0x249249u;
                const int cnt = __popc(nz);
                n_cand += cnt;
                n_ph[0] += __popc(f & 0x249249u);
                n_ph[1] += __popc((f >> 1) & 0x249249u);
                n_ph[2] += __popc((f >> 2) & 0x249249u);
                int total;
                int dst = ccount + scan().counts(cnt, total);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const uint32_t m = (f >> (3 * e)) & 7u;
                    if (m) s_cq[dst++] = (uint16_t) (((p0 + e) << 3) | m);
                }
                ccount += total;
                __syncthreads();
            }
            const long long tt2 = DBG_CLOCK();
            drain();
            const long long tt3 = DBG_CLOCK();
            while (vcount > 0) stage_b(vcount < kQuadsPerRound ? vcount : kQuadsPerRound);
            const long long tt4 = DBG_CLOCK();

            // ---- class bitmap of the tile + candidate class counters ----
            __syncthreads();
            if (tid >= kBlock - WAVE) {
                for (int w = tid - (kBlock - WAVE); w < kTile2 / 32; w += WAVE) {
                    const uint32_t uc = s_cls_uncond[w], cd = s_cls_cond[w] & ~uc;
                    p.class_bitmap[(D0 >> 5) + w] = cd;
                    n_cls_cond += __popc(cd);
                    n_cls_uncond += __popc(uc);
                }
            }
            if (tid == 0) { dbg_tile[0] += tt1 - tt0; dbg_tile[1] += tt2 - tt1; dbg_tile[2] += tt3 - tt2; dbg_tile[3] += tt4 - tt3; dbg_tile[4] += DBG_CLOCK() - tt4; }
        }
        flush();
        if (tid == kBlock - 1) p.unit_count[unit] = unit_records;
    }
    atomicAdd(&s_cnt[CNT_CANDIDATES], (unsigned long long) n_cand);
    atomicAdd(&s_cnt[CNT_PHASE0 + 0], (unsigned long long) n_ph[0]);
    atomicAdd(&s_cnt[CNT_PHASE0 + 2], (unsigned long long) n_ph[1]);
    atomicAdd(&s_cnt[CNT_PHASE0 + 4], (unsigned long long) n_ph[2]);
    atomicAdd(&s_cnt[CNT_CLASS_COND], (unsigned long long) n_cls_cond);
    atomicAdd(&s_cnt[CNT_CLASS_UNCOND], (unsigned long long) n_cls_uncond);
    atomicAdd(&s_cnt[CNT_RECORDS], (unsigned long long) n_rec);
    if (tid == 0) { s_cnt[10] = dbg_b; s_cnt[11] = dbg_nb; s_cnt[12] = dbg_take; s_cnt[13] = DBG_CLOCK() - dbg_t0; s_cnt[14] = dbg_s1; s_cnt[15] = dbg_s2; for (int k = 0; k < 6; ++k) s_cnt[CNT_DEBUG0 + k] = dbg_tile[k]; s_cnt[CNT_DEBUG0 + 6] = dbg_s3; }
    __syncthreads();
    if (tid < CNT_NUM && s_cnt[tid]) {
        unsigned long long v = s_cnt[tid];
        if (tid == CNT_PHASE0 + 0) { atomicAdd(&p.counters[CNT_PHASE0 + 0], v); atomicAdd(&p.counters[CNT_PHASE0 + 1], v); }
        else if (tid == CNT_PHASE0 + 2) { atomicAdd(&p.counters[CNT_PHASE0 + 2], v); atomicAdd(&p.counters[CNT_PHASE0 + 3], v); }
        else atomicAdd(&p.counters[tid], v);
    }
}

void launch_sweep_slice_v2(const SweepParams &p, hipStream_t s) {
    if (p.nunits == 0) return;
    // persistent grid = exactly the workgroups that are resident at once (occupancy x CUs): a larger
    // grid runs a ragged second wave of workgroups, a smaller one leaves CUs idle
    static int resident = 0;
    if (!resident) {
        int per_cu = 0, dev = 0, cus = 256;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
        const size_t dyn = (size_t) (p.n_long + p.n_short + 4) * sizeof(uint32_t);
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_sweep_slice_v2, kBlock, dyn) != hipSuccess || per_cu < 1) per_cu = 2;
        resident = per_cu * cus;
        if (resident > kSweepMaxBlocks) resident = kSweepMaxBlocks;
    }
    static unsigned env_blocks = 0;
    static bool env_read = false;
    if (!env_read) { env_read = true; if (const char *e = getenv("MGPU_SWEEP_BLOCKS")) env_blocks = (unsigned) atoi(e); }
    unsigned maxb = (unsigned) resident;
    if (env_blocks >= 1 && env_blocks < maxb) maxb = env_blocks;
    unsigned blocks = p.nunits < maxb ? p.nunits : maxb;
    hipLaunchKernelGGL(k_sweep_slice_v2, dim3(blocks), dim3(kBlock), (size_t) (p.n_long + p.n_short + 4) * sizeof(uint32_t), s, p);
}

// ---------------------------------------------------------------------------------------------
// k_sweep_slice (third generation): every WAVE is autonomous.  The second generation spent most
// of its time at workgroup barriers (~20 per tile, with 1-2 of 4 waves working in the slicing
// stages) — waves that never wait for each other let the CU interleave 16 independent instruction
// streams instead.  A wave owns a unit of the stream (kUnit positions, one record chain), walks it
// in wave-private LDS tiles of kWaveTile positions (+304 halo), and does everything itself with
// wave-level primitives (ballot / prefix sums, no __syncthreads after the table preload):
//   sweep 4 x 512 positions -> candidate queue -> pairs -> DF stage (lane = pair) -> ring of
//   valid pairs -> slicing (4 lanes per frame, 16 frames per round) -> CRC/score -> staged records
//   -> flush into the wave's reserved slice of the pool.  The next tile is prefetched into
//   registers while the current one is sliced.
// Shared by the workgroup (read-only after the preload, or benign races): group-syndrome tables,
// syndrome keys, the adder-address cache.
// ---------------------------------------------------------------------------------------------
// Register prefetch of the next tile costs 20 VGPRs and keeps the kernel at 3 waves/SIMD (<= 168 VGPRs).
// Measured alternative: -DMGPU_V3_PREFETCH=0 fits 4 waves/SIMD (128 VGPRs, no spills) but is 9 % slower
// (4.91 vs 4.52 ms per 537 M positions): the exposed tile-load latency costs more than the fourth wave hides.
#ifndef MGPU_V3_PREFETCH
#define MGPU_V3_PREFETCH 1
#endif
constexpr int kWT = kWaveTile;
constexpr int kWTChunks = (kWT + kHalo) / 8;                 // 16-byte chunks per tile (294)
constexpr int kWPre = (kWTChunks + WAVE - 1) / WAVE;         // 16-byte loads per lane and tile (5)
constexpr int kWStep = WAVE * 8;                             // positions per sweep step (512)
constexpr int kWCQCap = 3 * WAVE;                            // candidate queue (drained before it could overflow)
constexpr int kWPQCap = kWStep + WAVE;                       // pre-check survivor queue: one step's worth + the leftovers
constexpr int kWVCap = 128;                                  // ring of valid pairs (power of two)
constexpr int kWFrames = WAVE / 4;                           // frames sliced per round (16)
constexpr int kWStageCap = 24;                               // (generation 4) staged records per wave
constexpr int kWFrameCap = WAVE;                             // sliced frames waiting for the scoring pass
constexpr int kWAdderCache = 1024;

struct SlicedFrame {          // 32 bytes: a frame as sliced, waiting for CRC classification
    uint32_t W[4];            // frame bits 0..127, bit 0 = MSB of W[0]
    uint32_t synd;            // CRC-24 syndrome (112- or 56-bit, by DF)
    uint32_t meta;            // df | try-phase << 8
    uint32_t pos;             // scan position within the chunk
    uint32_t pad;
};

struct WaveLds {                                             // wave-private LDS, 16-byte aligned members first
    uint16_t mag[kWT + kHalo + 8];                           // 4720 B
    SlicedFrame frames[kWFrameCap];                          // 2048 B
    uint16_t cq[kWCQCap];                                    //  384 B
    uint16_t pq[kWPQCap];                                    // 1152 B  pre-check survivors waiting for the threshold tests
    uint16_t pairs[WAVE * 5];                                //  640 B
    uint32_t v[kWVCap];                                      //  512 B
};

#define WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)

// Everything scoreModesMessage / decodeModesMessage's CRC stage decide about one sliced frame without
// the ICAO filter (mode_s.c:276-419, 443-606): fills the two record halves, returns false when the
// frame scores -2 whatever the filter holds.
__device__ __forceinline__ bool classify_frame(const SweepParams &p, const uint32_t *s_gsyn, const uint32_t *s_keys,
                                               uint32_t W0, uint32_t W1, uint32_t W2, uint32_t W3, uint32_t synd, uint32_t df,
                                               int t, uint32_t pos, u32x4 &ra, u32x4 &rb, uint32_t &flags_out, uint32_t &addr_out) {
    const bool is_long = (p.valid_long >> df) & 1;
    const uint32_t aa = W0 & 0xffffffu;          // getbits(msg, 9, 32)
    int sk = -2, su = -2, fb0 = 0xff, fb1 = 0xff;
    uint32_t addr = 0, flags = is_long ? REC_LONG : 0;
    bool emit = false;
    if (is_long) {
        bool handled = false;
        if (p.fix_df && (df == 1 || df == 16 || df == 19 || df == 21 || df == 25)) {
            const int j = 4 - (__ffs(df ^ 17u) - 1);
            if (synd == s_gsyn[16u >> j]) {             // == bit_syndrome[j]; fixDF17msgtype, mode_s.c:276-301
                sk = 900; su = 700; addr = aa;
                flags |= REC_ACCEPT_IF_UNKNOWN | REC_DFFIX | (1u << REC_CORR_SHIFT);
                fb0 = j; emit = handled = true;
            }
        }
        if (!handled && !(W0 == 0 && (W1 >> 8) == 0)) {
            if (df == 16 || df == 20 || df == 21) {
                sk = 1000; su = -1; addr = synd; flags |= REC_COND; emit = true;      // Address/Parity
            } else if (df == 17 || df == 18) {
                int b0 = 0xff, b1 = 0xff;
                const int nerr = synd == 0 ? 0 : lane_diagnose(s_keys, p.tab_long, p.n_long, synd, b0, b1);
                if (nerr >= 0) {
                    uint32_t a2 = aa;
                    if (nerr >= 1) a2 = fix_aa(a2, b0);
                    if (nerr >= 2) a2 = fix_aa(a2, b1);
                    sk = 1800 / (nerr + 1); su = 1400 / (nerr + 1); addr = a2;
                    flags |= (uint32_t) nerr << REC_CORR_SHIFT;
                    if (a2 == aa) flags |= REC_ACCEPT_IF_UNKNOWN;   // mode_s.c:560: only a changed AA needs the filter
                    if (nerr == 0 && df == 17) flags |= REC_ADDER;
                    if (nerr >= 1) fb0 = b0;
                    if (nerr >= 2) fb1 = b1;
                    emit = true;
                }
            }
        }
    } else if (!(W0 == 0 && (W1 >> 8) == 0)) {
        if (df == 11) {
            if (synd & 0xffff80u) {
                int b0 = 0xff, b1 = 0xff;
                if (lane_diagnose(s_keys + p.n_long, p.tab_short, p.n_short, synd, b0, b1) == 1) {   // 2-bit errors are ambiguous in DF11
                    sk = 800; su = -1; addr = fix_aa(aa, b0);
                    flags |= REC_COND | (1u << REC_CORR_SHIFT);
                    fb0 = b0; emit = true;
                }
            } else if ((synd & 0x7f) == 0) {
                sk = 1600; su = 750; addr = aa; flags |= REC_ACCEPT_IF_UNKNOWN | REC_ADDER; emit = true;
            } else {
                sk = 1000; su = -1; addr = aa; flags |= REC_COND; emit = true;
            }
        } else {   // DF 0, 4, 5
            sk = 1000; su = -1; addr = synd; flags |= REC_COND; emit = true;
        }
    }
    if (!emit) return false;
    ra.x = pos;
    ra.y = (uint32_t) t | (flags << 8) | ((uint32_t) (uint16_t) sk << 16);
    ra.z = (uint32_t) (uint16_t) su | ((uint32_t) fb0 << 16) | ((uint32_t) fb1 << 24);
    ra.w = addr;
    if (!is_long) { W1 &= 0xffffff00u; W2 = 0; W3 = 0; }
    rb.x = __builtin_bswap32(W0); rb.y = __builtin_bswap32(W1);
    rb.z = __builtin_bswap32(W2); rb.w = __builtin_bswap32(W3) & 0xffffu;
    flags_out = flags;
    addr_out = addr;
    return true;
}

__global__ __launch_bounds__(kBlock, MGPU_V3_PREFETCH ? 3 : 4) void k_sweep_slice(SweepParams p) {
    __shared__ __attribute__((aligned(16))) WaveLds s_w[kBlock / WAVE];
    __shared__ uint32_t s_gsyn[(kGroupsLong + kGroupsShort) * 32];
    __shared__ uint32_t s_acache[kWAdderCache];
    __shared__ unsigned long long s_cnt[CNT_NUM];
    extern __shared__ uint32_t s_keys[];   // n_long + n_short sorted syndromes

    const int tid = threadIdx.x, lane = lane_id(), wv = tid >> 6;
    for (int i = tid; i < (kGroupsLong + kGroupsShort) * 32; i += kBlock) s_gsyn[i] = p.group_syndrome[i];
    for (int i = tid; i < kWAdderCache; i += kBlock) s_acache[i] = 0xFFFFFFFFu;
    for (int i = tid; i < p.n_long; i += kBlock) s_keys[i] = (uint32_t) (p.tab_long[i] >> 16);
    for (int i = tid; i < p.n_short; i += kBlock) s_keys[p.n_long + i] = (uint32_t) (p.tab_short[i] >> 16);
    if (tid < CNT_NUM) s_cnt[tid] = 0;
    __syncthreads();   // the only workgroup barrier before the final counter reduction

    WaveLds &L = s_w[wv];
    const uint32_t *w32 = (const uint32_t *) L.mag;
    const uint64_t lt_mask = (1ull << lane) - 1;
    const uint32_t wave_global = blockIdx.x * (kBlock / WAVE) + wv;
    const uint32_t nwaves = gridDim.x * (kBlock / WAVE);

    uint32_t n_cand = 0, n_ph[3] = {0, 0, 0}, n_rec = 0;
    long long tm[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // MGPU_KERNEL_TIMERS: load, sweep, stage A, slice, score, total, rounds B, passes
    const long long tm_start = DBG_CLOCK();
    (void) tm_start;
    // wave-uniform: reserved pool space.  The first slice is the wave's by position — a returning atomic on the shared
    // cursor costs ~11 ns serialised, and every wave of the grid asks for its first slice within the same few microseconds
    uint32_t chunk_base = wave_global * (uint32_t) kPoolChunkRecords, chunk_left = kPoolChunkRecords;
    u32x4 pre[kWPre];
    bool have_pre = false;

    for (uint32_t unit = wave_global; unit < p.nunits; unit += nwaves) {
        uint32_t prev_hdr = kNone, unit_records = 0;
        int fcount = 0;                               // sliced frames waiting in L.frames (wave-uniform)
        if (lane == 0) p.unit_first[unit] = kNone;

        // Scoring pass over the waiting frames, lane = frame: CRC classification, class planes, adder
        // bitmap, and the records go straight from registers into the wave's slice of the pool as one
        // segment of the unit's chain.  Runs right after a tile has been staged (never between a
        // prefetch and the wait for it), so its stores and atomics have a whole tile's time to retire
        // before the next s_waitcnt vmcnt(0).
        auto score_pass = [&]() __attribute__((always_inline)) {
            if (fcount == 0) return;
            if (p.debug_stage & 8) { fcount = 0; return; }
            const long long ts0 = DBG_CLOCK();
            bool emit = false;
            u32x4 ra = {0, 0, 0, 0}, rb = {0, 0, 0, 0};
            if (lane < fcount) {
                const u32x4 f0 = *(const u32x4 *) &L.frames[lane].W[0];
                const u32x4 f1 = *(const u32x4 *) &L.frames[lane].synd;
                uint32_t flags = 0, addr = 0;
                emit = classify_frame(p, s_gsyn, s_keys, f0.x, f0.y, f0.z, f0.w, f1.x, f1.y & 0xffu, (int) (f1.y >> 8), f1.z,
                                      ra, rb, flags, addr);
                if (emit) {
                    const uint32_t gpos = f1.z;
                    // class planes (zeroed per chunk); a 32-position word belongs to one unit = one wave, so
                    // workgroup scope suffices: the atomic runs in this XCD's L2, not at the memory side
                    if (p.debug_stage & 1) {} else if (flags & REC_COND) __hip_atomic_fetch_or(&p.class_bitmap[gpos >> 5], 1u << (gpos & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    else __hip_atomic_fetch_or(&p.class_uncond[gpos >> 5], 1u << (gpos & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if ((flags & REC_ADDER) && !(p.debug_stage & 2)) {
                        adder_publish<kWAdderCache>(p.adder_bitmap, s_acache, addr);
                    }
                }
            }
            const uint64_t em = __ballot(emit);
            const int cnt = __popcll(em);
            if (cnt && !(p.debug_stage & 4)) {
                if (chunk_left < (uint32_t) cnt + 1u) {
                    uint32_t b = 0;
                    if (lane == 0) b = atomicAdd(p.pool_used, (uint32_t) kPoolChunkRecords);
                    chunk_base = nwaves * (uint32_t) kPoolChunkRecords + rfl(b);     // the cursor counts from behind the waves' first slices
                    chunk_left = kPoolChunkRecords;
                }
                const uint32_t base = chunk_base;
                chunk_base += (uint32_t) cnt + 1u;
                chunk_left -= (uint32_t) cnt + 1u;
                if ((uint64_t) base + kPoolChunkRecords <= p.pool_cap) {
                    if (emit) {
                        u32x4 *d = (u32x4 *) &p.pool[base + 1 + __popcll(em & lt_mask)];
                        d[0] = ra; d[1] = rb;
                    }
                    if (lane == 0) {
                        u32x4 h0 = {(uint32_t) cnt, 0xFFu, 0u, kNone}, h1 = {0, 0, 0, 0};
                        u32x4 *hd = (u32x4 *) &p.pool[base];
                        hd[0] = h0; hd[1] = h1;
                        if (prev_hdr == kNone) p.unit_first[unit] = base; else p.pool[prev_hdr].addr = base;
                    }
                    prev_hdr = base;
                    unit_records += cnt;
                    n_rec += cnt;
                } else if (lane == 0) {
                    atomicAdd(&p.counters[CNT_POOL_OVERFLOW], 1ull);
                }
            }
            fcount = 0;
            WAVE_SYNC();
            tm[4] += DBG_CLOCK() - ts0; tm[7] += 1;
        };

        for (int tile = 0; tile < kUnit / kWT; ++tile) {
            const uint64_t D0 = (uint64_t) unit * kUnit + (uint64_t) tile * kWT;
            if (D0 >= p.n) break;
            // ---- tile into LDS (prefetched while the previous tile was sliced) ----
            const long long tl0 = DBG_CLOCK();
            if (!have_pre) {
#pragma unroll
                for (int k = 0; k < kWPre; ++k) {
                    const int i = lane + k * WAVE;
                    if (i < kWTChunks) pre[k] = *(const u32x4 *) &p.mag[D0 + 8 * i];
                }
            }
#pragma unroll
            for (int k = 0; k < kWPre; ++k) {
                const int i = lane + k * WAVE;
                if (i < kWTChunks) {   // the tile holds the samples with the top bit flipped: see slice_group_biased
                    const u32x4 v = {pre[k].x ^ 0x80008000u, pre[k].y ^ 0x80008000u, pre[k].z ^ 0x80008000u, pre[k].w ^ 0x80008000u};
                    *(u32x4 *) &L.mag[8 * i] = v;
                }
            }
            {
                uint64_t Dn = D0 + kWT;
                bool next_unit = false;
                if (tile + 1 >= kUnit / kWT || Dn >= p.n) { Dn = (uint64_t) (unit + nwaves) * kUnit; next_unit = true; }
                have_pre = MGPU_V3_PREFETCH && !(p.debug_stage & 16) && (next_unit ? (unit + nwaves < p.nunits) : true);
                if (have_pre) {
#pragma unroll
                    for (int k = 0; k < kWPre; ++k) {
                        const int i = lane + k * WAVE;
                        if (i < kWTChunks) pre[k] = *(const u32x4 *) &p.mag[Dn + 8 * i];
                    }
                }
            }
            WAVE_SYNC();
            tm[0] += DBG_CLOCK() - tl0;
            if (fcount > kWFrameCap - 2 * kWFrames) score_pass();   // typical place: stores overlap this tile's sweep

            int ccount = 0, vhead = 0, vcount = 0;

            // ---- slicing: 4 lanes per frame, 16 frames per round; the sliced frames wait in LDS ----
            auto stage_b = [&](int take) __attribute__((always_inline)) {
                if (fcount + take > kWFrameCap) score_pass();       // dense regions only
                const long long tb0 = DBG_CLOCK();
                const int quad = lane >> 2, qj = lane & 3;
                const bool have = quad < take;
                uint32_t e = 0, chunk = 0, psyn = 0, df = 0;
                bool is_long = false;
                int pos_local = 0, t = 4;
                if (have) {
                    e = L.v[(vhead + quad) & (kWVCap - 1)];
                    pos_local = (int) ((e & 0xffffu) >> 3);
                    t = 4 + (int) (e & 7u);
                    df = e >> 16;
                    is_long = (p.valid_long >> df) & 1;
                    if (is_long || qj < 2) {
                        SliceGeom g;
                        make_geom(pos_local, t, g);
                        slice_chunk<true>(w32, s_gsyn, g, is_long, qj, chunk, psyn);
                    }
                }
                const uint32_t c0 = quad_bcast<0>(chunk), c1 = quad_bcast<1>(chunk), c2 = quad_bcast<2>(chunk), c3 = quad_bcast<3>(chunk);
                const uint32_t sx = quad_bcast<0>(psyn) ^ quad_bcast<1>(psyn) ^ quad_bcast<2>(psyn) ^ quad_bcast<3>(psyn);
                if (have && qj == 0) {
                    u32x4 f0, f1;
                    f0.x = (df << 27) | (c0 >> 3);
                    f0.y = ((c0 & 7u) << 29) | (c1 >> 1);
                    f0.z = ((c1 & 1u) << 31) | (c2 << 1) | (c3 >> 19);
                    f0.w = (c3 << 13) & 0xffff0000u;
                    f1.x = sx ^ s_gsyn[(is_long ? 0 : kGroupsLong * 32) + df];
                    f1.y = df | ((uint32_t) t << 8);
                    f1.z = (uint32_t) (D0 + pos_local);
                    f1.w = 0;
                    SlicedFrame &fr = L.frames[fcount + quad];
                    *(u32x4 *) &fr.W[0] = f0;
                    *(u32x4 *) &fr.synd = f1;
                }
                fcount += take;
                vhead += take;
                vcount -= take;
                WAVE_SYNC();
                tm[3] += DBG_CLOCK() - tb0; tm[6] += 1;
            };

            // ---- candidates -> pairs -> DF stage -> ring ----
            auto drain = [&]() __attribute__((always_inline)) {
                for (int c0 = 0; c0 < ccount; c0 += WAVE) {
                    const int ci = c0 + lane;
                    const uint32_t code = ci < ccount ? L.cq[ci] : 0u;
                    const uint32_t mask = code & 7u;
                    const int np = 2 * (int) (mask & 1u) + (int) (mask & 2u) + (int) ((mask >> 2) & 1u);
                    int npairs;
                    int off = wave_excl_scan_small((uint32_t) np, npairs);
                    const uint32_t pl = code & 0xfff8u;
                    if (mask & 1u) { L.pairs[off++] = (uint16_t) (pl | 0u); L.pairs[off++] = (uint16_t) (pl | 1u); }
                    if (mask & 2u) { L.pairs[off++] = (uint16_t) (pl | 2u); L.pairs[off++] = (uint16_t) (pl | 3u); }
                    if (mask & 4u) { L.pairs[off++] = (uint16_t) (pl | 4u); }
                    WAVE_SYNC();
                    for (int a0 = 0; a0 < npairs; a0 += WAVE) {
                        const long long ta0 = DBG_CLOCK();
                        const int j = a0 + lane;
                        bool valid = false;
                        uint32_t entry = 0;
                        if (j < npairs) {
                            const uint32_t pc = L.pairs[j];
                            SliceGeom g;
                            make_geom((int) (pc >> 3), 4 + (int) (pc & 7u), g);
                            const uint32_t df = slice_group_biased(w32, g, 0);
                            valid = ((p.valid_long | p.valid_short) >> df) & 1;
                            entry = pc | (df << 16);
                        }
                        const uint64_t vm = __ballot(valid);
                        if (valid) L.v[(vhead + vcount + __popcll(vm & lt_mask)) & (kWVCap - 1)] = entry;
                        vcount += __popcll(vm);
                        WAVE_SYNC();
                        tm[2] += DBG_CLOCK() - ta0;
                        while (vcount >= kWFrames) stage_b(kWFrames);
                    }
                }
                ccount = 0;
            };

            // ---- threshold tests (demod_2400.c:324-378) for up to 64 pre-check survivors, lane = survivor ----
            // samples pa[1..18] as 9 dword pairs fetched at the survivor's own alignment
            int pqn = 0, pqh = 0;                                 // survivors waiting in L.pq[pqh .. pqn)
            auto eval_round = [&](int take) __attribute__((always_inline)) {
                if (ccount > kWCQCap - WAVE) { const long long td0 = DBG_CLOCK(); drain(); tm[1] -= DBG_CLOCK() - td0; }
                uint32_t m = 0;
                int pos = 0;
                if (lane < take) {
                    pos = L.pq[pqh + lane];
                    const int q = (pos + 1) >> 1;
                    const uint32_t sh = ((uint32_t) (pos + 1) & 1u) * 16u;
                    uint32_t d[10], P[9];
#pragma unroll
                    for (int k = 0; k < 10; ++k) d[k] = w32[q + k];
#pragma unroll
                    for (int k = 0; k < 9; ++k) P[k] = __builtin_amdgcn_alignbit(d[k + 1], d[k], sh) ^ 0x80008000u;   // (pa[2k+1], pa[2k+2]), bias removed
#define LO16(x) ((int) ((x) & 0xffffu))
#define HI16(x) ((int) ((x) >> 16))
                    const int m1 = LO16(P[0]), m2 = HI16(P[0]), m3 = LO16(P[1]), m4 = HI16(P[1]), m5 = LO16(P[2]), m8 = HI16(P[3]),
                              m9 = LO16(P[4]), m10 = HI16(P[4]), m11 = LO16(P[5]), m12 = HI16(P[5]), m16 = HI16(P[7]),
                              m17 = LO16(P[8]), m18 = HI16(P[8]);
#undef LO16
#undef HI16
                    const int base_noise = m5 + m8 + m16 + m17 + m18;
                    const int ref = (base_noise * p.thr) >> 5;
                    const int d23 = m2 - m3, s14 = m1 + m4, d1011 = m10 - m11;
                    const int common = s14 - d23 + m9 + m12;
                    if (common - d1011 >= ref) m |= 1;
                    if (common + d1011 >= ref) m |= 2;
                    if (s14 + 2 * d23 + d1011 + m12 >= ref) m |= 4;
                }
                const uint64_t cm = __ballot(m != 0);
                if (cm) {
                    n_cand += __popcll(cm);                       // wave-uniform tallies, added once per wave at the end
                    n_ph[0] += __popcll(__ballot(m & 1u));
                    n_ph[1] += __popcll(__ballot(m & 2u));
                    n_ph[2] += __popcll(__ballot(m & 4u));
                    if (m) L.cq[ccount + __popcll(cm & lt_mask)] = (uint16_t) ((pos << 3) | m);
                    ccount += __popcll(cm);
                    WAVE_SYNC();
                }
                pqh += take;
            };

            // ---- pre-check pa[1]>pa[7] && pa[12]>pa[14] && pa[12]>pa[15] (demod_2400.c:311-322), 512 positions per
            //      step, 8 per lane, two per packed-u16 instruction; about 1 position in 6 survives into L.pq ----
            const int nvalid = p.n - D0 < (uint64_t) kWT ? (int) (p.n - D0) : kWT;
            for (int sub = 0; sub < kWT / kWStep; ++sub) {
                if (sub * kWStep >= nvalid) break;
                const long long tw0 = DBG_CLOCK();
                const int p0 = sub * kWStep + lane * 8;
                uint32_t w[12];
                {
                    const u32x4 a = *(const u32x4 *) &L.mag[p0], b = *(const u32x4 *) &L.mag[p0 + 8],
                                c = *(const u32x4 *) &L.mag[p0 + 16];
                    w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w;
                    w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
                    w[8] = c.x; w[9] = c.y; w[10] = c.z; w[11] = c.w;
                }
                uint32_t odd[11];                                  // odd[k] = (m[2k+1], m[2k+2]), m[i] = sample p0+i
#pragma unroll
                for (int k = 0; k < 11; ++k) odd[k] = __builtin_amdgcn_alignbit(w[k + 1], w[k], 16);
                uint32_t sm = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) {                      // positions p0+2j (low half) and p0+2j+1 (high half)
                    // (biased samples: unsigned order of the magnitudes = signed order of what the tile holds)
                    const v2i16 A = __builtin_bit_cast(v2i16, odd[j]), B = __builtin_bit_cast(v2i16, odd[j + 3]),
                                C = __builtin_bit_cast(v2i16, w[j + 6]), Dd = __builtin_bit_cast(v2i16, w[j + 7]),
                                E = __builtin_bit_cast(v2i16, odd[j + 7]);
                    const v2i16 t1 = __builtin_elementwise_sub_sat(A, B);                                 // > 0 <=> pa[1] > pa[7]
                    const v2i16 t2 = __builtin_elementwise_sub_sat(C, __builtin_elementwise_max(Dd, E));  // > 0 <=> pa[12] > max(pa[14], pa[15])
                    const v2i16 r = __builtin_elementwise_min(t1, t2);
                    if (r.x > 0) sm |= 1u << (2 * j);
                    if (r.y > 0) sm |= 2u << (2 * j);
                }
                {   // positions beyond the end of the stream (last tile only)
                    const int room = nvalid - p0;
                    if (room < 8) sm = room <= 0 ? 0u : (sm & ((1u << room) - 1u));
                }
                const int cnt = __popc(sm);
                int total;
                int dst = pqn + wave_excl_scan_small((uint32_t) cnt, total);
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if ((sm >> e) & 1u) L.pq[dst++] = (uint16_t) (p0 + e);
                pqn += total;
                WAVE_SYNC();
                while (pqn - pqh >= WAVE) eval_round(WAVE);
                if (pqh) {                                         // keep the (< 64) leftovers at the front
                    const int rem = pqn - pqh;
                    uint16_t t = 0;
                    if (lane < rem) t = L.pq[pqh + lane];
                    WAVE_SYNC();
                    if (lane < rem) L.pq[lane] = t;
                    pqn = rem; pqh = 0;
                    WAVE_SYNC();
                }
                tm[1] += DBG_CLOCK() - tw0;
            }
            if (pqn) eval_round(pqn);
            drain();
            while (vcount > 0) stage_b(vcount < kWFrames ? vcount : kWFrames);
        }
        score_pass();
        if (lane == 0) p.unit_count[unit] = unit_records;
    }
    if (lane == 0) {
        atomicAdd(&s_cnt[CNT_CANDIDATES], (unsigned long long) n_cand);
        atomicAdd(&s_cnt[CNT_PHASE0 + 0], (unsigned long long) n_ph[0]);
        atomicAdd(&s_cnt[CNT_PHASE0 + 2], (unsigned long long) n_ph[1]);
        atomicAdd(&s_cnt[CNT_PHASE0 + 4], (unsigned long long) n_ph[2]);
        atomicAdd(&s_cnt[CNT_RECORDS], (unsigned long long) n_rec);
    }
#if MGPU_KERNEL_TIMERS
    tm[5] = DBG_CLOCK() - tm_start;
    if (lane == 0) for (int i = 0; i < 8; ++i) atomicAdd(&s_cnt[CNT_DEBUG0 + i], (unsigned long long) tm[i]);
#endif
    __syncthreads();
    if (tid < CNT_NUM && s_cnt[tid]) {
        unsigned long long v = s_cnt[tid];
        if (tid == CNT_PHASE0 + 0) { atomicAdd(&p.counters[CNT_PHASE0 + 0], v); atomicAdd(&p.counters[CNT_PHASE0 + 1], v); }
        else if (tid == CNT_PHASE0 + 2) { atomicAdd(&p.counters[CNT_PHASE0 + 2], v); atomicAdd(&p.counters[CNT_PHASE0 + 3], v); }
        else atomicAdd(&p.counters[tid], v);
    }
}


// persistent grid of generation 3 = the workgroups resident at once (occupancy x CUs) for this much dynamic LDS
// (the 2-bit syndrome keys of --aggressive take 20 KB: 2 workgroups per CU instead of 3)
static unsigned sweep_slice_grid(size_t dyn) {
    static size_t cached_dyn = ~(size_t) 0;
    static unsigned cached = 0;
    if (dyn != cached_dyn) {
        int per_cu = 0, dev = 0, cus = 256;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_sweep_slice, kBlock, dyn) != hipSuccess || per_cu < 1) per_cu = 2;
        unsigned resident = (unsigned) (per_cu * cus);
        if (resident > (unsigned) kSweepMaxBlocks) resident = kSweepMaxBlocks;
        static unsigned env_blocks = 0;
        static bool env_read = false;
        if (!env_read) { env_read = true; if (const char *e = getenv("MGPU_SWEEP_BLOCKS")) env_blocks = (unsigned) atoi(e); }
        if (env_blocks >= 1 && env_blocks < resident) resident = env_blocks;
        cached = resident;
        cached_dyn = dyn;
    }
    return cached;
}

void launch_sweep_slice(const SweepParams &p, hipStream_t s) {
    if (p.nunits == 0) return;
    const size_t dyn = (size_t) (p.n_long + p.n_short + 4) * sizeof(uint32_t);
    const unsigned maxb = sweep_slice_grid(dyn);
    const unsigned want = (p.nunits + (kBlock / WAVE) - 1) / (kBlock / WAVE);
    const unsigned blocks = want < maxb ? want : maxb;
    hipLaunchKernelGGL(k_sweep_slice, dim3(blocks), dim3(kBlock), dyn, s, p);
}

// ---------------------------------------------------------------------------------------------
// Generation 4: the sweep and the slicer are separate kernels.
//
//   k_sweep  pure streaming: every magnitude is read once (16-byte loads, no LDS), each lane evaluates
//            8 consecutive positions from a 26-sample register window and the wave appends the
//            candidates of its unit, in position order, to the unit's slice of `cand`.  This is the
//            kernel the HBM roofline applies to: 2 bytes per position in, ~1 % of positions out.
//   k_slice  one wave per unit again, but it only touches the samples frames actually need
//            (~1 % of positions x 5 rows x 2 dwords for the DF stage, 22 groups x 11 dwords for the
//            ~0.4 % that survive it), read straight from global memory: the chunk's magnitudes
//            (134 MB) were just streamed and sit in the 256 MB Infinity Cache.  No LDS tile, so the
//            occupancy is set by registers alone and the slicer's latency no longer stalls the sweep.
// ---------------------------------------------------------------------------------------------
constexpr int kSwStep = WAVE * 8;        // positions per wave step

__global__ __launch_bounds__(kBlock) void k_sweep(SweepParams p) {
    __shared__ unsigned long long s_cnt[8];
    const int tid = threadIdx.x, lane = lane_id();
    if (tid < 8) s_cnt[tid] = 0;
    __syncthreads();
    const uint32_t wave_global = blockIdx.x * (kBlock / WAVE) + (tid >> 6);
    const uint32_t nwaves = gridDim.x * (kBlock / WAVE);
    uint32_t n_cand = 0, n_ph[3] = {0, 0, 0};

    for (uint32_t unit = wave_global; unit < p.nunits; unit += nwaves) {
        const uint64_t U0 = (uint64_t) unit * kUnit;
        uint16_t *out = p.cand + U0;
        int count = 0;
        u32x4 a, b, c;
        uint32_t d;
        {
            const uint16_t *src = p.mag + U0 + lane * 8;
            a = *(const u32x4 *) src; b = *(const u32x4 *) (src + 8); c = *(const u32x4 *) (src + 16); d = *(const uint32_t *) (src + 24);
        }
        for (int step = 0; step < kUnit / kSwStep; ++step) {
            const uint64_t S0 = U0 + (uint64_t) step * kSwStep;
            if (S0 >= p.n) break;
            uint32_t w[13] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w, d};
            if (step + 1 < kUnit / kSwStep && S0 + kSwStep < p.n) {      // next step's window while this one is evaluated
                const uint16_t *src = p.mag + S0 + kSwStep + lane * 8;
                a = *(const u32x4 *) src; b = *(const u32x4 *) (src + 8); c = *(const u32x4 *) (src + 16); d = *(const uint32_t *) (src + 24);
            }
            const uint64_t P0 = S0 + (uint64_t) lane * 8;
            uint32_t f = 0;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
#define SM(i) ((int) ((w[((e) + (i)) >> 1] >> ((((e) + (i)) & 1) * 16)) & 0xffffu))
                const bool pc = SM(1) > SM(7) && SM(12) > SM(14) && SM(12) > SM(15);
                const int base_noise = SM(5) + SM(8) + SM(16) + SM(17) + SM(18);
                const int ref = (base_noise * p.thr) >> 5;
                const int d23 = SM(2) - SM(3), s14 = SM(1) + SM(4), d1011 = SM(10) - SM(11);
                const int common = s14 - d23 + SM(9) + SM(12);
                uint32_t m = 0;
                if (common - d1011 >= ref) m |= 1;
                if (common + d1011 >= ref) m |= 2;
                if (s14 + 2 * d23 + d1011 + SM(12) >= ref) m |= 4;
#undef SM
                if (!pc || P0 + e >= p.n) m = 0;
                f |= m << (3 * e);
            }
            const uint32_t nz = (f | (f >> 1) | (f >> 2)) & 0x249249u;
            const int cnt = __popc(nz);
            n_cand += cnt;
            n_ph[0] += __popc(f & 0x249249u);
            n_ph[1] += __popc((f >> 1) & 0x249249u);
            n_ph[2] += __popc((f >> 2) & 0x249249u);
            if (__ballot(cnt != 0)) {
                int total;
                int dst = count + wave_excl_scan(cnt, total);
                const int pl0 = step * kSwStep + lane * 8;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const uint32_t m = (f >> (3 * e)) & 7u;
                    if (m) out[dst++] = (uint16_t) (((pl0 + e) << 3) | m);
                }
                count += total;
            }
        }
        if (lane == 0) p.cand_count[unit] = (uint32_t) count;
    }
    atomicAdd(&s_cnt[0], (unsigned long long) n_cand);
    atomicAdd(&s_cnt[1], (unsigned long long) n_ph[0]);
    atomicAdd(&s_cnt[2], (unsigned long long) n_ph[1]);
    atomicAdd(&s_cnt[3], (unsigned long long) n_ph[2]);
    __syncthreads();
    if (tid == 0 && s_cnt[0]) atomicAdd(&p.counters[CNT_CANDIDATES], s_cnt[0]);
    if (tid == 1 && s_cnt[1]) { atomicAdd(&p.counters[CNT_PHASE0 + 0], s_cnt[1]); atomicAdd(&p.counters[CNT_PHASE0 + 1], s_cnt[1]); }
    if (tid == 2 && s_cnt[2]) { atomicAdd(&p.counters[CNT_PHASE0 + 2], s_cnt[2]); atomicAdd(&p.counters[CNT_PHASE0 + 3], s_cnt[2]); }
    if (tid == 3 && s_cnt[3]) atomicAdd(&p.counters[CNT_PHASE0 + 4], s_cnt[3]);
}

struct SliceLds {                                            // wave-private LDS of k_slice
    PhaseRec stage[kWStageCap];
    uint16_t pairs[WAVE * 5];
    uint32_t v[kWVCap];
};

__global__ __launch_bounds__(kBlock) void k_slice(SweepParams p) {
    __shared__ __attribute__((aligned(16))) SliceLds s_w[kBlock / WAVE];
    __shared__ uint32_t s_gsyn[(kGroupsLong + kGroupsShort) * 32];
    __shared__ uint32_t s_acache[kWAdderCache];
    __shared__ unsigned long long s_cnt[4];
    extern __shared__ uint32_t s_keys[];

    const int tid = threadIdx.x, lane = lane_id(), wv = tid >> 6;
    for (int i = tid; i < (kGroupsLong + kGroupsShort) * 32; i += kBlock) s_gsyn[i] = p.group_syndrome[i];
    for (int i = tid; i < kWAdderCache; i += kBlock) s_acache[i] = 0xFFFFFFFFu;
    for (int i = tid; i < p.n_long; i += kBlock) s_keys[i] = (uint32_t) (p.tab_long[i] >> 16);
    for (int i = tid; i < p.n_short; i += kBlock) s_keys[p.n_long + i] = (uint32_t) (p.tab_short[i] >> 16);
    if (tid < 4) s_cnt[tid] = 0;
    __syncthreads();

    SliceLds &L = s_w[wv];
    const uint64_t lt_mask = (1ull << lane) - 1;
    const uint32_t wave_global = blockIdx.x * (kBlock / WAVE) + wv;
    const uint32_t nwaves = gridDim.x * (kBlock / WAVE);
    uint32_t n_cls_cond = 0, n_cls_uncond = 0, n_rec = 0;
    uint32_t chunk_base = 0, chunk_left = 0;

    for (uint32_t unit = wave_global; unit < p.nunits; unit += nwaves) {
        const uint64_t U0 = (uint64_t) unit * kUnit;
        const uint32_t *w32 = (const uint32_t *) (p.mag + U0);      // unit-relative sample pairs, global memory
        const uint16_t *cq = p.cand + U0;
        const int ccount = (int) p.cand_count[unit];
        uint32_t prev_hdr = kNone, unit_records = 0;
        int scount = 0, vhead = 0, vcount = 0;
        if (lane == 0) p.unit_first[unit] = kNone;
        uint32_t last_cond_pos = 0xFFFFFFFFu, last_uncond_pos = 0xFFFFFFFFu;   // lane-local: last position this lane classified

        auto flush = [&]() __attribute__((always_inline)) {
            if (scount == 0) return;
            if (chunk_left < (uint32_t) scount + 1u) {
                uint32_t b = 0;
                if (lane == 0) b = atomicAdd(p.pool_used, (uint32_t) kPoolChunkRecords);
                chunk_base = rfl(b);
                chunk_left = kPoolChunkRecords;
            }
            const uint32_t base = chunk_base;
            chunk_base += (uint32_t) scount + 1u;
            chunk_left -= (uint32_t) scount + 1u;
            if ((uint64_t) base + kPoolChunkRecords <= p.pool_cap) {
                if (lane < scount) {
                    const u32x4 *src = (const u32x4 *) &L.stage[lane];
                    u32x4 *d = (u32x4 *) &p.pool[base + 1 + lane];
                    d[0] = src[0]; d[1] = src[1];
                }
                if (lane == 0) {
                    u32x4 h0 = {(uint32_t) scount, 0xFFu, 0u, kNone}, h1 = {0, 0, 0, 0};
                    u32x4 *hd = (u32x4 *) &p.pool[base];
                    hd[0] = h0; hd[1] = h1;
                    if (prev_hdr == kNone) p.unit_first[unit] = base; else p.pool[prev_hdr].addr = base;
                }
                prev_hdr = base;
                unit_records += scount;
                n_rec += scount;
            } else if (lane == 0) {
                atomicAdd(&p.counters[CNT_POOL_OVERFLOW], 1ull);
            }
            scount = 0;
            WAVE_SYNC();
        };

        auto stage_b = [&](int take) __attribute__((always_inline)) {
            const int quad = lane >> 2, qj = lane & 3;
            const bool have = quad < take;
            uint32_t e = 0, chunk = 0, psyn = 0, df = 0;
            bool is_long = false;
            int pos_local = 0, t = 4;
            if (have) {
                e = L.v[(vhead + quad) & (kWVCap - 1)];
                pos_local = (int) ((e & 0xffffu) >> 3);
                t = 4 + (int) (e & 7u);
                df = e >> 16;
                is_long = (p.valid_long >> df) & 1;
                if (is_long || qj < 2) {
                    SliceGeom g;
                    make_geom(pos_local, t, g);
                    slice_chunk(w32, s_gsyn, g, is_long, qj, chunk, psyn);
                }
            }
            const uint32_t c0 = quad_bcast<0>(chunk), c1 = quad_bcast<1>(chunk), c2 = quad_bcast<2>(chunk), c3 = quad_bcast<3>(chunk);
            const uint32_t sx = quad_bcast<0>(psyn) ^ quad_bcast<1>(psyn) ^ quad_bcast<2>(psyn) ^ quad_bcast<3>(psyn);
            bool emit = false;
            u32x4 ra = {0, 0, 0, 0}, rb = {0, 0, 0, 0};
            if (have && qj == 0) {
                uint32_t W[4];
                W[0] = (df << 27) | (c0 >> 3);
                W[1] = ((c0 & 7u) << 29) | (c1 >> 1);
                W[2] = ((c1 & 1u) << 31) | (c2 << 1) | (c3 >> 19);
                W[3] = (c3 << 13) & 0xffff0000u;
                const uint32_t synd = sx ^ s_gsyn[(is_long ? 0 : kGroupsLong * 32) + df];
                const uint32_t aa = W[0] & 0xffffffu;
                int sk = -2, su = -2, fb0 = 0xff, fb1 = 0xff;
                uint32_t addr = 0, flags = is_long ? REC_LONG : 0;
                if (is_long) {
                    bool handled = false;
                    if (p.fix_df && (df == 1 || df == 16 || df == 19 || df == 21 || df == 25)) {
                        const int j = 4 - (__ffs(df ^ 17u) - 1);
                        if (synd == s_gsyn[16u >> j]) {
                            sk = 900; su = 700; addr = aa;
                            flags |= REC_ACCEPT_IF_UNKNOWN | REC_DFFIX | (1u << REC_CORR_SHIFT);
                            fb0 = j; emit = handled = true;
                        }
                    }
                    if (!handled && !(W[0] == 0 && (W[1] >> 8) == 0)) {
                        if (df == 16 || df == 20 || df == 21) {
                            sk = 1000; su = -1; addr = synd; flags |= REC_COND; emit = true;
                        } else if (df == 17 || df == 18) {
                            int b0 = 0xff, b1 = 0xff;
                            const int nerr = synd == 0 ? 0 : lane_diagnose(s_keys, p.tab_long, p.n_long, synd, b0, b1);
                            if (nerr >= 0) {
                                uint32_t a2 = aa;
                                if (nerr >= 1) a2 = fix_aa(a2, b0);
                                if (nerr >= 2) a2 = fix_aa(a2, b1);
                                sk = 1800 / (nerr + 1); su = 1400 / (nerr + 1); addr = a2;
                                flags |= (uint32_t) nerr << REC_CORR_SHIFT;
                                if (a2 == aa) flags |= REC_ACCEPT_IF_UNKNOWN;
                                if (nerr == 0 && df == 17) flags |= REC_ADDER;
                                if (nerr >= 1) fb0 = b0;
                                if (nerr >= 2) fb1 = b1;
                                emit = true;
                            }
                        }
                    }
                } else if (!(W[0] == 0 && (W[1] >> 8) == 0)) {
                    if (df == 11) {
                        if (synd & 0xffff80u) {
                            int b0 = 0xff, b1 = 0xff;
                            if (lane_diagnose(s_keys + p.n_long, p.tab_short, p.n_short, synd, b0, b1) == 1) {
                                sk = 800; su = -1; addr = fix_aa(aa, b0);
                                flags |= REC_COND | (1u << REC_CORR_SHIFT);
                                fb0 = b0; emit = true;
                            }
                        } else if ((synd & 0x7f) == 0) {
                            sk = 1600; su = 750; addr = aa; flags |= REC_ACCEPT_IF_UNKNOWN | REC_ADDER; emit = true;
                        } else {
                            sk = 1000; su = -1; addr = aa; flags |= REC_COND; emit = true;
                        }
                    } else {
                        sk = 1000; su = -1; addr = synd; flags |= REC_COND; emit = true;
                    }
                }
                if (emit) {
                    const uint64_t gpos = U0 + (uint64_t) pos_local;
                    ra.x = (uint32_t) gpos;
                    ra.y = (uint32_t) t | (flags << 8) | ((uint32_t) (uint16_t) sk << 16);
                    ra.z = (uint32_t) (uint16_t) su | ((uint32_t) fb0 << 16) | ((uint32_t) fb1 << 24);
                    ra.w = addr;
                    if (!is_long) { W[1] &= 0xffffff00u; W[2] = 0; W[3] = 0; }
                    rb.x = __builtin_bswap32(W[0]); rb.y = __builtin_bswap32(W[1]);
                    rb.z = __builtin_bswap32(W[2]); rb.w = __builtin_bswap32(W[3]) & 0xffffu;
                    // class planes (zeroed per chunk): bit = candidate has a conditional / an unconditional record
                    // (a 32-position word belongs to exactly one unit = one wave, so workgroup scope is enough: the
                    //  atomic runs in this XCD's L2, not at the memory side)
                    if (flags & REC_COND) {
                        if (last_cond_pos != (uint32_t) gpos) { __hip_atomic_fetch_or(&p.class_bitmap[gpos >> 5], 1u << (gpos & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); last_cond_pos = (uint32_t) gpos; }
                    } else if (last_uncond_pos != (uint32_t) gpos) {
                        __hip_atomic_fetch_or(&p.class_uncond[gpos >> 5], 1u << (gpos & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); last_uncond_pos = (uint32_t) gpos;
                    }
                    if (flags & REC_ADDER) {
                        adder_publish<kWAdderCache>(p.adder_bitmap, s_acache, addr);
                    }
                }
            }
            const uint64_t em = __ballot(emit);
            if (emit) {
                u32x4 *d = (u32x4 *) &L.stage[scount + __popcll(em & lt_mask)];
                d[0] = ra; d[1] = rb;
            }
            scount += __popcll(em);
            vhead += take;
            vcount -= take;
            WAVE_SYNC();
            if (scount > kWStageCap - kWFrames) flush();
        };

        for (int c0 = 0; c0 < ccount; c0 += WAVE) {
            const int ci = c0 + lane;
            const uint32_t code = ci < ccount ? cq[ci] : 0u;
            const uint32_t mask = code & 7u;
            const int np = 2 * (int) (mask & 1u) + (int) (mask & 2u) + (int) ((mask >> 2) & 1u);
            int npairs;
            int off = wave_excl_scan(np, npairs);
            const uint32_t pl = code & 0xfff8u;
            if (mask & 1u) { L.pairs[off++] = (uint16_t) (pl | 0u); L.pairs[off++] = (uint16_t) (pl | 1u); }
            if (mask & 2u) { L.pairs[off++] = (uint16_t) (pl | 2u); L.pairs[off++] = (uint16_t) (pl | 3u); }
            if (mask & 4u) { L.pairs[off++] = (uint16_t) (pl | 4u); }
            WAVE_SYNC();
            for (int a0 = 0; a0 < npairs; a0 += WAVE) {
                const int j = a0 + lane;
                bool valid = false;
                uint32_t entry = 0;
                if (j < npairs) {
                    const uint32_t pc = L.pairs[j];
                    SliceGeom g;
                    make_geom((int) (pc >> 3), 4 + (int) (pc & 7u), g);
                    const uint32_t df = slice_group(w32, g, 0);
                    valid = ((p.valid_long | p.valid_short) >> df) & 1;
                    entry = pc | (df << 16);
                }
                const uint64_t vm = __ballot(valid);
                if (valid) L.v[(vhead + vcount + __popcll(vm & lt_mask)) & (kWVCap - 1)] = entry;
                vcount += __popcll(vm);
                WAVE_SYNC();
                while (vcount >= kWFrames) stage_b(kWFrames);
            }
        }
        while (vcount > 0) stage_b(vcount < kWFrames ? vcount : kWFrames);
        flush();
        if (lane == 0) p.unit_count[unit] = unit_records;
    }
    if (lane == 0) atomicAdd(&s_cnt[0], (unsigned long long) n_rec);
    (void) n_cls_cond; (void) n_cls_uncond;
    __syncthreads();
    if (tid == 0 && s_cnt[0]) atomicAdd(&p.counters[CNT_RECORDS], s_cnt[0]);
}

// Class planes -> final class bitmap (cond & ~uncond) + the two class counters; one pass over n/32 words.
// The planes are handed back zeroed, ready for the slot's next chunk (no memset on the stream).
__device__ __forceinline__ void class_finalize_part(uint32_t block, uint32_t nblocks, uint32_t *cond, uint32_t *uncond, uint32_t *final_bitmap,
                                                    uint64_t nwords, unsigned long long *counters) {
    __shared__ unsigned long long s_c[2];
    if (threadIdx.x < 2) s_c[threadIdx.x] = 0;
    __syncthreads();
    uint32_t nc = 0, nu = 0;
    const uint64_t nvec = (nwords + 3) / 4;      // the planes are allocated (and zeroed) in whole 16-byte groups
    const u32x4 zero = {0, 0, 0, 0};
    for (uint64_t i = (uint64_t) block * kBlock + threadIdx.x; i < nvec; i += (uint64_t) nblocks * kBlock) {
        const u32x4 uc = ((const u32x4 *) uncond)[i], c0 = ((const u32x4 *) cond)[i];
        const u32x4 cd = {c0.x & ~uc.x, c0.y & ~uc.y, c0.z & ~uc.z, c0.w & ~uc.w};
        ((u32x4 *) final_bitmap)[i] = cd;
        if (c0.x | c0.y | c0.z | c0.w) ((u32x4 *) cond)[i] = zero;
        if (uc.x | uc.y | uc.z | uc.w) ((u32x4 *) uncond)[i] = zero;
        nc += __popc(cd.x) + __popc(cd.y) + __popc(cd.z) + __popc(cd.w);
        nu += __popc(uc.x) + __popc(uc.y) + __popc(uc.z) + __popc(uc.w);
    }
    nc = (uint32_t) wave_sum_u64(nc);
    nu = (uint32_t) wave_sum_u64(nu);
    if (lane_id() == 0) { atomicAdd(&s_c[0], (unsigned long long) nc); atomicAdd(&s_c[1], (unsigned long long) nu); }
    __syncthreads();
    if (threadIdx.x == 0 && s_c[0]) atomicAdd(&counters[CNT_CLASS_COND], s_c[0]);
    if (threadIdx.x == 1 && s_c[1]) atomicAdd(&counters[CNT_CLASS_UNCOND], s_c[1]);
}

static int resident_blocks(const void *kernel, size_t dyn_lds) {
    int per_cu = 0, dev = 0, cus = 256;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, kBlock, dyn_lds) != hipSuccess || per_cu < 1) per_cu = 2;
    if (per_cu > 8) per_cu = 8;
    return per_cu * cus;
}

void launch_sweep(const SweepParams &p, hipStream_t s) {
    if (p.nunits == 0) return;
    static int resident = 0;
    if (!resident) resident = resident_blocks((const void *) k_sweep, 0);
    const unsigned want = (p.nunits + (kBlock / WAVE) - 1) / (kBlock / WAVE);
    const unsigned blocks = want < (unsigned) resident ? want : (unsigned) resident;
    hipLaunchKernelGGL(k_sweep, dim3(blocks), dim3(kBlock), 0, s, p);
}

void launch_slice(const SweepParams &p, hipStream_t s) {
    if (p.nunits == 0) return;
    const size_t dyn = (size_t) (p.n_long + p.n_short + 4) * sizeof(uint32_t);
    static int resident = 0;
    if (!resident) resident = resident_blocks((const void *) k_slice, dyn);
    const unsigned want = (p.nunits + (kBlock / WAVE) - 1) / (kBlock / WAVE);
    const unsigned blocks = want < (unsigned) resident ? want : (unsigned) resident;
    hipLaunchKernelGGL(k_slice, dim3(blocks), dim3(kBlock), dyn, s, p);
}

// =============================================================================================
// pre-screen: a conditional record (score_unknown < 0) can only matter if its address is one
// that some clean DF17 / DF11-IID0 frame of this stream carries (the only frames that ever add
// to the ICAO filter, mode_s.c:766-779).  Everything else is "rejected_unknown_icao" for sure.
// =============================================================================================

__device__ __forceinline__ bool rec_live(const PhaseRec &r, const uint32_t *bitmap) {
    if (!(r.flags & REC_COND)) return true;
    const uint32_t a = r.addr & 0xffffffu;
    return (bitmap[a >> 5] >> (a & 31)) & 1;
}

// WRITE pass: besides compacting the live records (straight into pinned host memory) the wave also
// computes, for every live record, the signal power the reference would report if this record
// became the accepted frame: sum of mag^2 over d_mag[pos+19 .. pos+19+len), len = 268 / 134 by the
// DF as sliced (demod_2400.c:399,436-457).  ~5 records per real frame, 5 coalesced loads per lane
// each — and the ordered walk then needs no second GPU round trip.
// MODE 0: COUNT pass (decides which records live; with `keep_masks` it leaves the decision as a 64-bit mask in
//         each segment header's spare bytes — generation 3 segments hold at most 64 records);
// MODE 1: WRITE pass that decides again (same inputs, same stream: same answer);
// MODE 2: WRITE pass that reads the masks — it may then run beside the next chunk's sweep, which adds bits
//         to the adder bitmap (a second look at the bitmap could disagree with the counted offsets).
template <int MODE>
__device__ __forceinline__ uint32_t prescreen_unit(uint32_t u, PhaseRec *pool, const uint32_t *unit_first, uint32_t nunits,
                                                   const uint32_t *bitmap, uint32_t dst0, PhaseRec *live,
                                                   const uint16_t *mag, unsigned long long *live_sig, bool keep_masks,
                                                   unsigned long long *counters) {
    constexpr bool WRITE = MODE != 0;
    const int lane = lane_id();
    if (u >= nunits) return 0;
    uint32_t h = unit_first[u];
    uint32_t nlive = 0;
    while (h != kNone) {
        const uint32_t cnt = pool[h].pos, next = pool[h].addr;
        for (uint32_t i0 = 0; i0 < cnt; i0 += WAVE) {
            const uint32_t i = i0 + lane;
            bool ok = false;
            uint32_t pos = 0, len = 0;
            uint64_t m;
            if (MODE == 2) {
                m = ((const unsigned long long *) &pool[h])[2];
                ok = (m >> lane) & 1;
                if (ok) {
                    const PhaseRec &r = pool[h + 1 + i];
                    pos = r.pos;
                    len = (r.msg[0] & 0x80) ? 268u : 134u;
                }
            } else {
                uint32_t addr = 0;
                int sk = 0, su = 0;
                if (i < cnt) {
                    const PhaseRec &r = pool[h + 1 + i];
                    ok = rec_live(r, bitmap);
                    pos = r.pos;
                    addr = r.addr;
                    sk = r.score_known;
                    su = r.score_unknown;
                    len = (r.msg[0] & 0x80) ? 268u : 134u;
                }
                // Dominated records: an earlier try-phase of the same position with the same address and scores at
                // least as good wins every comparison the walk can make (same address = same filter answer, the
                // best-phase test is a strict '>', demod_2400.c:246) — typically 2 of the 3 records of a clean frame.
                const uint64_t live0 = __ballot(ok);
                bool dom = false;
#pragma unroll
                for (int d = 1; d <= 4; ++d) {
                    const uint32_t pj = __shfl_up(pos, d), aj = __shfl_up(addr, d);
                    const int kj = __shfl_up(sk, d), uj = __shfl_up(su, d);
                    if (lane >= d && ((live0 >> (lane - d)) & 1) && pj == pos && aj == addr && kj >= sk && uj >= su) dom = true;
                }
                ok = ok && !dom;
                m = __ballot(ok);
                if (MODE == 0 && keep_masks && lane == 0) {
                    if (cnt > (uint32_t) WAVE) atomicAdd(&counters[CNT_POOL_OVERFLOW], 1ull);   // cannot happen: one scoring pass = one segment
                    ((unsigned long long *) &pool[h])[2] = m;
                }
            }
            if (WRITE) {
                if (ok) {
                    const uint32_t d = dst0 + nlive + __popcll(m & ((1ull << lane) - 1));
                    const u32x4 *src = (const u32x4 *) &pool[h + 1 + i];
                    u32x4 *dd = (u32x4 *) &live[d];
                    dd[0] = src[0]; dd[1] = src[1];
                }
                uint64_t todo = m;
                uint32_t k = 0;
                while (todo) {
                    const int src_lane = __ffsll((unsigned long long) todo) - 1;
                    todo &= todo - 1;
                    const uint32_t p0 = __builtin_amdgcn_readlane(pos, src_lane);
                    const uint32_t n = __builtin_amdgcn_readlane(len, src_lane);
                    const uint16_t *sm = mag + p0 + 19;
                    unsigned long long acc = 0;
                    for (uint32_t q = lane; q < n; q += WAVE) { const uint32_t v = sm[q]; acc += (unsigned long long) (v * v); }
                    acc = wave_sum_u64(acc);
                    if (lane == 0) live_sig[dst0 + nlive + k] = acc;
                    ++k;
                }
            }
            nlive += __popcll(m);
        }
        h = next;
    }
    return nlive;
}

// COUNT pass (one wave per unit) and, in the remaining workgroups of the same launch, the class-plane
// finalize: two small latency-bound jobs that do not depend on each other.
__global__ __launch_bounds__(kBlock) void k_count_finalize(PhaseRec *pool, const uint32_t *unit_first, uint32_t nunits,
                                                           const uint32_t *bitmap, uint32_t *unit_live, uint32_t *block_live, uint32_t nb_count,
                                                           uint32_t *cond, uint32_t *uncond, uint32_t *final_bitmap, uint64_t nwords,
                                                           unsigned long long *counters, int keep_masks) {
    if (blockIdx.x < nb_count) {
        __shared__ uint32_t s_live[kBlock / WAVE];
        const uint32_t wv = threadIdx.x >> 6, u = blockIdx.x * (kBlock / WAVE) + wv;
        const uint32_t nlive = prescreen_unit<0>(u, pool, unit_first, nunits, bitmap, 0, nullptr, nullptr, nullptr, keep_masks != 0, counters);
        if (lane_id() == 0) { s_live[wv] = nlive; if (u < nunits) unit_live[u] = nlive; }
        __syncthreads();
        if (threadIdx.x == 0) block_live[blockIdx.x] = s_live[0] + s_live[1] + s_live[2] + s_live[3];   // the write pass sums these: no scan kernel
    } else {
        class_finalize_part(blockIdx.x - nb_count, gridDim.x - nb_count, cond, uncond, final_bitmap, nwords, counters);
    }
}

// WRITE pass.  A workgroup's output offset = the live counts of all workgroups before it (block_live, at most a few
// thousand words out of L2, summed cooperatively) + those of the earlier units of its own four.
template <int MODE>
__global__ __launch_bounds__(kBlock) void k_prescreen_write(PhaseRec *pool, const uint32_t *unit_first, uint32_t nunits,
                                                            const uint32_t *bitmap, const uint32_t *unit_live, const uint32_t *block_live,
                                                            PhaseRec *live, const uint16_t *mag, unsigned long long *live_sig,
                                                            unsigned long long *counters) {
    __shared__ uint32_t s_part[kBlock / WAVE];
    uint32_t acc = 0;
    for (uint32_t i = threadIdx.x; i < blockIdx.x; i += kBlock) acc += block_live[i];
    acc = (uint32_t) wave_sum_u64(acc);
    const uint32_t wv = threadIdx.x >> 6;
    if (lane_id() == 0) s_part[wv] = acc;
    __syncthreads();
    uint32_t dst0 = s_part[0] + s_part[1] + s_part[2] + s_part[3];
    const uint32_t u = blockIdx.x * (kBlock / WAVE) + wv;
    for (uint32_t v = blockIdx.x * (kBlock / WAVE); v < u && v < nunits; ++v) dst0 += unit_live[v];
    const uint32_t nlive = prescreen_unit<MODE>(u, pool, unit_first, nunits, bitmap, dst0, live, mag, live_sig, false, nullptr);
    if (u == nunits - 1 && lane_id() == 0) counters[CNT_LIVE_TOTAL] = dst0 + nlive;
}

// The chunk's scratch block (counters, pool cursor, per-buffer sums) goes to the host's pinned copy and is
// handed back zeroed for the slot's next chunk: replaces a D2H copy and a memset on the stream.
__global__ __launch_bounds__(kBlock) void k_publish(unsigned long long *d_scratch, unsigned long long *h_scratch, uint32_t nwords) {
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < nwords; i += gridDim.x * kBlock) {
        h_scratch[i] = d_scratch[i];
        d_scratch[i] = 0;
    }
}

// count + finalize on `s`; write + publish on `s_write` (== s, or a second stream when the segment headers carry
// the live masks: `q.keep_masks`), ordered after the count pass by `ev_scan`
int launch_prescreen(const PostSweepParams &q, hipStream_t s, hipStream_t s_write, hipEvent_t ev_scan) {
    if (q.nunits == 0) return 0;
    const unsigned nb_count = (q.nunits + 3) / 4;
    unsigned nb_fin = 0;
    if (q.class_final) {
        nb_fin = (unsigned) ((q.class_words / 4 + kBlock) / kBlock);
        if (nb_fin > 256) nb_fin = 256;          // every workgroup ends with two device atomics on the same two words
    }
    hipLaunchKernelGGL(k_count_finalize, dim3(nb_count + nb_fin), dim3(kBlock), 0, s, q.pool, q.unit_first, q.nunits, q.adder_bitmap,
                       q.unit_live, q.block_live, nb_count, q.class_cond, q.class_uncond, q.class_final, q.class_words, q.counters,
                       q.keep_masks ? 1 : 0);
    if (s_write != s) {
        if (hipEventRecord(ev_scan, s) != hipSuccess || hipStreamWaitEvent(s_write, ev_scan, 0) != hipSuccess) return -1;
    }
    if (q.keep_masks)
        hipLaunchKernelGGL(k_prescreen_write<2>, dim3(nb_count), dim3(kBlock), 0, s_write, q.pool, q.unit_first, q.nunits, q.adder_bitmap,
                           q.unit_live, q.block_live, q.live, q.mag, q.live_sig, q.counters);
    else
        hipLaunchKernelGGL(k_prescreen_write<1>, dim3(nb_count), dim3(kBlock), 0, s_write, q.pool, q.unit_first, q.nunits, q.adder_bitmap,
                           q.unit_live, q.block_live, q.live, q.mag, q.live_sig, q.counters);
    unsigned pb = (q.scratch_words + kBlock - 1) / kBlock;
    if (pb > 64) pb = 64;
    hipLaunchKernelGGL(k_publish, dim3(pb), dim3(kBlock), 0, s_write, q.d_scratch, q.h_scratch, q.scratch_words);
    return 0;
}

// =============================================================================================
// Mode A/C replies (demodulate2400AC, demod_2400.c:575-761): F1/F2 framing pulses 20.3 us apart,
// 20 bit slots of 1.45 us = 87 cycles of a virtual 60 MHz clock; one 2.4 MHz sample = 25 cycles.
// Position-parallel: every position that passes all of the loop body's tests becomes a candidate;
// the reference's skip over an accepted reply (:765) is applied in order on the host.
// =============================================================================================

// per buffer: noise_level = (mean_power + sqrt(mean_power - mean_level^2)) * 65535 + 0.5 (:579-580), from the
// converter's sums with the converter's own divisions (convert.c:101-107)
__global__ __launch_bounds__(kBlock) void k_modeac_noise(const unsigned long long *sum_level, const unsigned long long *sum_power,
                                                         const double *fsum_level, const double *fsum_power, int format,
                                                         uint64_t n, uint32_t B, uint32_t nbuf, uint32_t *noise_level) {
#pragma clang fp contract(off)
    const uint32_t b = blockIdx.x * kBlock + threadIdx.x;
    if (b >= nbuf) return;
    const uint64_t first = (uint64_t) b * B;
    const double len = (double) (uint32_t) (n - first < B ? n - first : B);
    double ml, mp;
    if (format == 0) { ml = (double) sum_level[b] / 65536.0 / len; mp = (double) sum_power[b] / 65535.0 / 65535.0 / len; }
    else { ml = fsum_level[b] / len; mp = fsum_power[b] / len; }
    const double sd = __dsqrt_rn(mp - ml * ml);
    noise_level[b] = (uint32_t) ((mp + sd) * 65535 + 0.5);
}

// everything after the cheap F1 edge / level tests, for the rare lanes that get here
__device__ __forceinline__ bool modeac_try(const uint16_t *m /* the buffer's data[] */, uint32_t f1_sample, uint32_t m0, uint32_t m1,
                                           uint32_t f1_level, uint32_t noise_level, uint32_t pos, AcCand &cand) {
#pragma clang fp contract(off)
    // initial clock phase from the power that ended up in the second sample (:655-658): float arithmetic, then + 0.5 in double
    const float f1a_power = (float) m0 * (float) m0;
    const float f1b_power = (float) m1 * (float) m1;
    const float fraction = __fdiv_rn(f1b_power, f1a_power + f1b_power);
    const float at = (float) f1_sample + fraction * fraction;
    const uint32_t f1_clock = (uint32_t) ((double) (25.0f * at) + 0.5);
    const uint32_t f2_clock = f1_clock + 87 * 14;
    const uint32_t f2_sample = f2_clock / 25;
    if (!(m[f2_sample - 1] < m[f2_sample + 0])) return false;                                       // :666
    if (m[f2_sample + 2] > m[f2_sample + 0] || m[f2_sample + 2] > m[f2_sample + 1]) return false;  // :669
    const uint32_t f2_level = ((uint32_t) m[f2_sample + 0] + m[f2_sample + 1]) / 2;
    if (noise_level * 2 > f2_level) return false;
    const uint32_t f1f2_level = f1_level > f2_level ? f1_level : f2_level;
    const float midpoint = __fsqrt_rn((float) (noise_level * f1f2_level));                      // :683: unsigned product, then float
    const uint32_t signal_threshold = (uint32_t) ((double) midpoint * 1.41421356237309504880 + 0.5);        // +3 dB
    const uint32_t noise_threshold = (uint32_t) (__ddiv_rn((double) midpoint, 1.41421356237309504880) + 0.5);   // -3 dB
    uint32_t bits = 0, bad = 0, clock = f1_clock;
    for (int bit = 0; bit < 20; ++bit, clock += 87) {                                          // :692-713
        const uint32_t sample = clock / 25;
        const uint32_t a = m[sample + 0], b = m[sample + 1], c = m[sample + 2];
        bits <<= 1;
        if (c >= signal_threshold) bad = 1;                                                    // noisy quiet period
        if (a >= signal_threshold || b >= signal_threshold) bits |= 1;
        else if (a > noise_threshold && b > noise_threshold) bad = 1;                          // uncertain
    }
    if ((bits & 0x80020u) != 0x80020u || (bits & 0x0101Bu) != 0 || bad) return false;               // :716-727
    const uint32_t modeac =
        ((bits & 0x40000) ? 0x0010 : 0) | ((bits & 0x20000) ? 0x1000 : 0) | ((bits & 0x10000) ? 0x0020 : 0) |
        ((bits & 0x08000) ? 0x2000 : 0) | ((bits & 0x04000) ? 0x0040 : 0) | ((bits & 0x02000) ? 0x4000 : 0) |
        ((bits & 0x00800) ? 0x0100 : 0) | ((bits & 0x00400) ? 0x0001 : 0) | ((bits & 0x00200) ? 0x0200 : 0) |
        ((bits & 0x00100) ? 0x0002 : 0) | ((bits & 0x00080) ? 0x0400 : 0) | ((bits & 0x00040) ? 0x0004 : 0) |
        ((bits & 0x00004) ? 0x0080 : 0);
    cand.pos = pos; cand.f2_clock = f2_clock; cand.modeac = modeac;
    return true;
}

// A workgroup covers 2048 positions; its samples (+ 80 of look-ahead: F2 is 48.7 samples after F1, the last bit slot
// 66) are staged in LDS once.  The cheap F1 tests leave a few dozen positions per workgroup (mostly pulses of Mode S
// frames); they are gathered in LDS and then tried one per lane, so that the long slow path — some 60 dependent,
// scattered sample reads — runs converged and out of LDS.
constexpr int kAcTile = kBlock * 8, kAcHalo = 80, kAcQueue = 1024;
__global__ __launch_bounds__(kBlock) void k_modeac(const uint16_t *mag, uint64_t n, uint32_t B, const uint32_t *noise_level,
                                                   AcCand *out, uint32_t cap, unsigned long long *list_counts, unsigned long long *counters) {
    __shared__ __attribute__((aligned(16))) uint16_t s_m[8 + kAcTile + kAcHalo];    // s_m[8 + i] = sample blk0 + i
    __shared__ uint32_t s_n;
    __shared__ uint16_t s_pos[kAcQueue];
    const uint64_t blk0 = (uint64_t) blockIdx.x * kAcTile;
    if (threadIdx.x == 0) s_n = 0;
    for (int i = threadIdx.x; i < (8 + kAcTile + kAcHalo) / 8; i += kBlock) {
        const int64_t g = (int64_t) blk0 - 8 + 8 * i;                          // d_mag is padded beyond n + 326 (api.cpp: alloc_slot)
        u32x4 v = {0, 0, 0, 0};
        if (g >= 0) v = *(const u32x4 *) &mag[g];
        *(u32x4 *) &s_m[8 * i] = v;
    }
    __syncthreads();
    const uint32_t b = (uint32_t) (blk0 / B);                                 // 2048 | B: the workgroup's positions share a buffer
    const uint64_t first = (uint64_t) b * B;
    const uint32_t nl = noise_level[b];
    const int l0 = 8 + threadIdx.x * 8;                                       // index of the thread's first position in s_m
    const uint64_t p0 = blk0 + (uint64_t) threadIdx.x * 8;
    if (p0 < n) {
        const u32x4 x = *(const u32x4 *) &s_m[l0];
        const uint32_t prev = *(const uint32_t *) &s_m[l0 - 2], next = *(const uint32_t *) &s_m[l0 + 8];
        const uint32_t w[6] = {prev, x.x, x.y, x.z, x.w, next};              // samples p0-2 .. p0+9
#pragma unroll
        for (int e = 0; e < 8; ++e) {
#define SMP(i) ((w[((e) + (i) + 2) >> 1] >> ((((e) + (i)) & 1) * 16)) & 0xffffu)
            const uint32_t m_1 = SMP(-1), m0 = SMP(0), m1 = SMP(1), m2 = SMP(2);
#undef SMP
            const uint64_t D = p0 + e;
            const uint32_t f1_sample = (uint32_t) (D - first);
            if (D >= n || f1_sample == 0) continue;                           // the loop starts at f1_sample = 1 (:582)
            if (!(m_1 < m0)) continue;                                        // not a rising edge (:639)
            if (m2 > m0 || m2 > m1) continue;                                 // quiet part not quiet (:642)
            if (nl * 2 > (m0 + m1) / 2) continue;                             // 6 dB above noise (:647)
            const uint32_t slot = atomicAdd(&s_n, 1u);
            if (slot < (uint32_t) kAcQueue) s_pos[slot] = (uint16_t) (threadIdx.x * 8 + e);
        }
    }
    __syncthreads();
    const uint32_t cnt = s_n;
    if (cnt > (uint32_t) kAcQueue) {              // half the positions passing is not a real signal; say so rather than drop any
        if (threadIdx.x == 0) atomicAdd(&counters[CNT_POOL_OVERFLOW], 1ull);
        return;
    }
    // the buffer's data[] as the slow path indexes it (f1_sample-relative), served from the LDS window
    const uint16_t *m_rel = s_m + 8 - (int64_t) (blk0 - first);
    __shared__ uint32_t s_nacc, s_base;
    __shared__ AcCand s_acc[kAcQueue];
    if (threadIdx.x == 0) s_nacc = 0;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < cnt; i += kBlock) {
        const uint32_t q = s_pos[i];
        const uint32_t m0 = s_m[8 + q], m1 = s_m[8 + q + 1];
        AcCand c;
        if (modeac_try(m_rel, (uint32_t) (blk0 - first) + q, m0, m1, (m0 + m1) / 2, nl, (uint32_t) (blk0 + q), c)) s_acc[atomicAdd(&s_nacc, 1u)] = c;
    }
    __syncthreads();
    // one returning device atomic per workgroup that found something (one per candidate on a single word costs 11 ns
    // each, serialised: more than the whole scan)
    const uint32_t nacc = s_nacc;
    if (nacc == 0) return;
    // ... and spread over kAcLists lists, each with its own counter word and its own slice of the output
    const uint32_t list = blockIdx.x % kAcLists, cap_l = cap / kAcLists;
    if (threadIdx.x == 0) s_base = (uint32_t) atomicAdd(&list_counts[list], (unsigned long long) nacc);
    __syncthreads();
    const uint32_t base = s_base;
    for (uint32_t i = threadIdx.x; i < nacc; i += kBlock)
        if (base + i < cap_l) out[(size_t) list * cap_l + base + i] = s_acc[i];
}

void launch_modeac(const uint16_t *mag, uint64_t n, uint32_t buf_samples, int format, const unsigned long long *sum_level,
                   const unsigned long long *sum_power, const double *fsum_level, const double *fsum_power,
                   uint32_t *noise_level, AcCand *out, uint32_t cap, unsigned long long *list_counts, unsigned long long *counters,
                   hipStream_t s) {
    if (n == 0) return;
    const uint32_t nbuf = (uint32_t) ((n + buf_samples - 1) / buf_samples);
    hipLaunchKernelGGL(k_modeac_noise, dim3((nbuf + kBlock - 1) / kBlock), dim3(kBlock), 0, s, sum_level, sum_power, fsum_level, fsum_power,
                       format, n, buf_samples, nbuf, noise_level);
    const uint64_t threads = (n + 7) / 8;
    hipLaunchKernelGGL(k_modeac, dim3((unsigned) ((threads + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, mag, n, buf_samples, noise_level,
                       out, cap, list_counts, counters);
}

// =============================================================================================
// per accepted message: signal power, and what its skip-ahead window hid from the counters
// =============================================================================================

__global__ __launch_bounds__(kBlock) void k_signal_power(const uint16_t *mag, const uint32_t *pos, const uint16_t *len,
                                                         uint32_t nmsg, unsigned long long *out) {
    const int lane = lane_id();
    const uint32_t wave_global = (blockIdx.x * kBlock + threadIdx.x) >> 6;
    const uint32_t nwaves = (gridDim.x * kBlock) >> 6;
    for (uint32_t i = wave_global; i < nmsg; i += nwaves) {
        const uint16_t *s = mag + pos[i] + 19;
        const int n = len[i];
        unsigned long long acc = 0;
        for (int k = lane; k < n; k += WAVE) { const uint32_t v = s[k]; acc += (unsigned long long) (v * v); }
        acc = wave_sum_u64(acc);
        if (lane == 0) out[i] = acc;
    }
}

void launch_signal_power(const uint16_t *mag, const uint32_t *pos, const uint16_t *len, uint32_t nmsg,
                         unsigned long long *out, hipStream_t s) {
    if (nmsg == 0) return;
    unsigned blocks = (nmsg + 3) / 4;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_signal_power, dim3(blocks), dim3(kBlock), 0, s, mag, pos, len, nmsg, out);
}

// The reference never looks at the `skip` positions after an accepted frame (pa += msglen*2,
// demod_2400.c:468), so candidates there count neither as preambles nor as rejects.  The sweep
// counted every candidate; this kernel re-evaluates the threshold tests on each window
// (<= 224 positions) and totals what has to be subtracted.
__global__ __launch_bounds__(kBlock) void k_window_stats(const uint16_t *mag, uint64_t n, int thr, const uint32_t *class_bitmap,
                                                         const uint32_t *pos, const uint16_t *skip, const uint32_t *limit,
                                                         uint32_t nmsg, unsigned long long *out) {
    __shared__ unsigned long long s_acc[8];
    if (threadIdx.x < 8) s_acc[threadIdx.x] = 0;
    __syncthreads();
    const int lane = lane_id();
    const uint32_t wave_global = (blockIdx.x * kBlock + threadIdx.x) >> 6;
    const uint32_t nwaves = (gridDim.x * kBlock) >> 6;
    uint32_t c_cand = 0, c_a = 0, c_b = 0, c_c = 0, c_cond = 0;
    for (uint32_t i = wave_global; i < nmsg; i += nwaves) {
        const uint32_t first = pos[i] + 1;
        uint32_t last = pos[i] + skip[i];            // inclusive
        if (last >= limit[i]) last = limit[i] - 1;   // the walk restarts at every buffer boundary
        for (uint32_t q = first + lane; q <= last; q += WAVE) {
            if (q >= n) break;
            const uint16_t *pa = mag + q;
            uint32_t m = 0;
            if (pa[1] > pa[7] && pa[12] > pa[14] && pa[12] > pa[15]) {
                const int base_noise = pa[5] + pa[8] + pa[16] + pa[17] + pa[18];
                const int ref = (base_noise * thr) >> 5;
                const int d23 = pa[2] - pa[3], s14 = pa[1] + pa[4], d1011 = pa[10] - pa[11];
                const int common = s14 - d23 + pa[9] + pa[12];
                if (common - d1011 >= ref) m |= 1;
                if (common + d1011 >= ref) m |= 2;
                if (s14 + 2 * d23 + d1011 + pa[12] >= ref) m |= 4;
            }
            if (m) {
                ++c_cand;
                c_a += m & 1; c_b += (m >> 1) & 1; c_c += (m >> 2) & 1;
                c_cond += (class_bitmap[q >> 5] >> (q & 31)) & 1;
            }
        }
    }
    atomicAdd(&s_acc[0], (unsigned long long) c_cand);
    atomicAdd(&s_acc[1], (unsigned long long) c_a);
    atomicAdd(&s_acc[2], (unsigned long long) c_b);
    atomicAdd(&s_acc[3], (unsigned long long) c_c);
    atomicAdd(&s_acc[4], (unsigned long long) c_cond);
    __syncthreads();
    if (threadIdx.x < 5 && s_acc[threadIdx.x]) atomicAdd(&out[threadIdx.x], s_acc[threadIdx.x]);
}

void launch_window_stats(const uint16_t *mag, uint64_t n, int thr, const uint32_t *class_bitmap, const uint32_t *pos,
                         const uint16_t *skip, const uint32_t *limit, uint32_t nmsg, unsigned long long *out, hipStream_t s) {
    if (nmsg == 0) return;
    unsigned blocks = (nmsg + 3) / 4;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(k_window_stats, dim3(blocks), dim3(kBlock), 0, s, mag, n, thr, class_bitmap, pos, skip, limit, nmsg, out);
}

}  // namespace mgpu
