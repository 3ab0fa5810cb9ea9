// kernels.hip — hand-written HIP kernels (gfx950 / CDNA4, wave64) for readsb's Mode-S hot path.
//
//   k_convert_*      IQ -> u16 magnitude                   (convert.c:64-108, 212-250, 329-367)
//   k_sweep          preamble sweep over every sample position: packed pre-check, wave prefix-sum compaction of
//                    the survivors, threshold tests -> ordered candidate lists     (demod_2400.c:290-378)
//   k_slice          PPM bit slicing of each tried phase straight from an LDS-staged sample window,
//                    wave-parallel CRC-24, syndrome lookup and the filter-independent half of
//                    scoreModesMessage                     (demod_2400.c:74-93,133-258; mode_s.c:276-419;
//                                                           crc.c:67-82,383-406)
//   k_prescreen_*    drops records that can only ever score "unknown ICAO"
//   k_window_stats   what the skip-ahead hid from the counters (demod_2400.c:468)
//   k_modeac*, k_beast_*, k_decode_fields: Mode A/C demodulator, beast wire encoder, per-message field decode
//
// No MFMA anywhere: this is HBM-bound integer/byte streaming work.  All arithmetic on the
// message path is integer and bit-exact with the reference; the SC16 converters use IEEE float
// ops with contraction disabled and a correctly rounded sqrt.
// One translation unit, in parts (kernels/*.inc, included below in dependency order).  -DMGPU_EXPERIMENTS (make exp ->
// libmodes_gpu_exp.so) adds the ordered walk on the device and the A/B switches of DESIGN.md §7.
#include "kernels.h"
#include "tables.h"

#include <atomic>
#include <cstdlib>

namespace mgpu {

#define WAVE 64

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
typedef uint32_t u32x4_a8 __attribute__((ext_vector_type(4), aligned(8)));
typedef uint32_t u32x2_a8 __attribute__((ext_vector_type(2), aligned(8)));
typedef unsigned short v2u16 __attribute__((ext_vector_type(2)));
typedef short v2i16 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int lane_id() { return threadIdx.x & (WAVE - 1); }

__device__ __forceinline__ uint32_t rfl(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }

// inclusive prefix sum over the wave, DPP only: row_shr 1/2/4/8 inside the rows of 16, then row_bcast 15 / 31
// across them (the sequence LLVM's atomic optimizer emits for gfx9).  total = lane 63's value.
__device__ __forceinline__ int wave_incl_scan_dpp(int v, int &total) {
    int x = v;
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);    // row_shr:1
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);    // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);    // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);    // row_shr:8
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);    // row_bcast:15 -> rows 1, 3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);    // row_bcast:31 -> rows 2, 3
    total = __builtin_amdgcn_readlane(x, WAVE - 1);
    return x;
}


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

// The dealers' request (k_sweep's steps, k_slice's tiles): one returning device atomic from lane 0 whose answer is read later in the
// SAME iteration of the caller's loop, into a variable declared inside that iteration.  Two things the compiler otherwise does to
// such a request, both of which put an s_waitcnt vmcnt(0) right behind the atomic (rounds 3-4 shipped k_slice that way: every
// request waited for at once, and the tile's prefetched samples with it): on a uniform address LLVM's atomic optimizer rewrites it
// into ballot + atomic + v_readfirstlane of the result — hence the offset through a VGPR it cannot see through; and a result
// variable that lives across iterations is redefined conditionally, which costs a v_mov of the in-flight register at the merge.
// (`opaque_zero`: made once per kernel by deal_opaque_zero() — made at every request, its register is the ticket's of the request
// before, and the compiler waits for that one first.)
__device__ __forceinline__ uint32_t deal_opaque_zero() {
    uint32_t off = 0;
    asm volatile("" : "+v"(off));
    return off;
}
__device__ __forceinline__ uint32_t deal_ask(uint32_t *counter, uint32_t opaque_zero) {   // call from ONE lane
    return __hip_atomic_fetch_add(counter + opaque_zero, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    return v;
}

#include "kernels/convert.inc"
#include "kernels/slicer.inc"
#include "kernels/sweep.inc"
#include "kernels/slice.inc"
#include "kernels/prescreen.inc"
#include "kernels/modeac.inc"
#include "kernels/window_stats.inc"
#include "kernels/build.inc"
#if MGPU_EXPERIMENTS
#include "kernels/walk.inc"      // the ordered walk on the device: experiments build only (the host owns the walk)
#endif
#include "kernels/beast.inc"
#include "kernels/fields.inc"
#include "kernels/gate.inc"

}  // namespace mgpu
