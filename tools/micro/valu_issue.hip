// micro-benchmark: VALU issue rate on gfx950 for the instruction classes k_sweep_slice is made of.
//   hipcc --offload-arch=gfx950 -O3 -o valu_issue valu_issue.hip && ./valu_issue
// For each class: a loop of 64 independent instructions (8 accumulators x 8), run by 1, 2, 4 and 8 waves
// per SIMD on every CU.  Reported: wave-instructions per shader clock per SIMD (1/2 = one wave64
// instruction every two clocks = all 32 lanes of a SIMD-32 busy; 1/4 = half rate) and the chip-wide
// lane-op rate.  The shader clock is measured with s_memtime inside the same kernels.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

// one asm statement per accumulator; a[i] is read and written, b and c are loop-invariant inputs
#define DEF_KERNEL(NAME, ASM)                                                                       \
    __global__ __launch_bounds__(256) void NAME(int iters, uint32_t *out, unsigned long long *clk) { \
        uint32_t a[8], b = threadIdx.x * 2654435761u + 12345u, c = blockIdx.x * 40503u + 977u;       \
        for (int i = 0; i < 8; ++i) a[i] = b ^ (i * 0x9e3779b9u);                                     \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                   \
        for (int it = 0; it < iters; ++it) {                                                          \
            _Pragma("unroll") for (int r = 0; r < 8; ++r) {                                           \
                _Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile(ASM : "+v"(a[i]) : "v"(b), "v"(c) : "vcc", "s20", "s21", "s22", "s23"); \
            }                                                                                         \
        }                                                                                             \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                   \
        uint32_t x = 0;                                                                               \
        for (int i = 0; i < 8; ++i) x ^= a[i];                                                        \
        if (x == 0x12345678u) out[threadIdx.x] = x;                                                   \
        if (threadIdx.x == 0 && blockIdx.x == 0) *clk = t1 - t0;                                      \
    }

DEF_KERNEL(k_add_u32, "v_add_u32 %0, %0, %1")
DEF_KERNEL(k_xor, "v_xor_b32 %0, %0, %1")
DEF_KERNEL(k_and_or, "v_and_or_b32 %0, %0, %1, %2")
DEF_KERNEL(k_lshl_or, "v_lshl_or_b32 %0, %0, 5, %1")
DEF_KERNEL(k_alignbit, "v_alignbit_b32 %0, %0, %1, %2")
DEF_KERNEL(k_bfe, "v_bfe_u32 %0, %0, 3, 16")
DEF_KERNEL(k_pk_sub_i16, "v_pk_sub_i16 %0, %0, %1 clamp")
DEF_KERNEL(k_pk_max_i16, "v_pk_max_i16 %0, %0, %1")
DEF_KERNEL(k_pk_add_u16, "v_pk_add_u16 %0, %0, %1")
DEF_KERNEL(k_dot2_i16, "v_dot2_i32_i16 %0, %1, %2, %0")
DEF_KERNEL(k_mul_lo, "v_mul_lo_u32 %0, %0, %1")
DEF_KERNEL(k_mul_u24, "v_mul_u32_u24 %0, %0, %1")
DEF_KERNEL(k_mad_u24, "v_mad_u32_u24 %0, %0, %1, %2")
DEF_KERNEL(k_mbcnt, "v_mbcnt_lo_u32_b32 %0, %1, %0")
DEF_KERNEL(k_cmp_cndmask, "v_cmp_gt_i32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc")
DEF_KERNEL(k_cmp_sgpr, "v_cmp_gt_i32 s[20:21], %0, %1\n s_and_b64 s[22:23], s[20:21], exec\n v_add_u32 %0, %0, %2")
DEF_KERNEL(k_fma_f32, "v_fma_f32 %0, %0, %1, %2")
DEF_KERNEL(k_pk_fma_f32, "v_fmac_f32 %0, %1, %2")
DEF_KERNEL(k_dpp_mov, "v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
DEF_KERNEL(k_readlane, "v_readlane_b32 s20, %0, 3\n v_add_u32 %0, s20, %0")
DEF_KERNEL(k_sdwa, "v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_0")

typedef void (*kern_t)(int, uint32_t *, unsigned long long *);
struct Entry { const char *name; kern_t k; int per_stmt; };

int main(int argc, char **argv) {
    Entry es[] = {
        {"v_add_u32", k_add_u32, 1}, {"v_xor_b32", k_xor, 1}, {"v_and_or_b32", k_and_or, 1}, {"v_lshl_or_b32", k_lshl_or, 1},
        {"v_alignbit_b32", k_alignbit, 1}, {"v_bfe_u32", k_bfe, 1}, {"v_pk_sub_i16 clamp", k_pk_sub_i16, 1},
        {"v_pk_max_i16", k_pk_max_i16, 1}, {"v_pk_add_u16", k_pk_add_u16, 1}, {"v_dot2_i32_i16", k_dot2_i16, 1},
        {"v_mul_lo_u32", k_mul_lo, 1}, {"v_mul_u32_u24", k_mul_u24, 1}, {"v_mad_u32_u24", k_mad_u24, 1},
        {"v_mbcnt_lo", k_mbcnt, 1}, {"v_cmp+v_cndmask (vcc)", k_cmp_cndmask, 2}, {"v_cmp->sgpr + s_and + v_add", k_cmp_sgpr, 2},
        {"v_fma_f32", k_fma_f32, 1}, {"v_fmac_f32", k_pk_fma_f32, 1}, {"v_mov_b32 dpp quad_perm", k_dpp_mov, 1},
        {"v_readlane + v_add", k_readlane, 2}, {"v_add_u32 sdwa", k_sdwa, 1},
    };
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    uint32_t *out; unsigned long long *clk, hclk = 0;
    hipMalloc(&out, 4096); hipMalloc(&clk, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    printf("# %s, %d CUs; loop body = 64 statements, %d iterations; columns: waves per SIMD\n", prop.name, cus, iters);
    printf("# value = VALU wave-instructions per shader clock per SIMD (0.50 = SIMD-32 full rate, 0.25 = half rate)\n");
    printf("%-30s %8s %8s %8s %8s   %s\n", "instruction", "1", "2", "4", "8", "clk MHz / chip T lane-ops/s at 8");
    FILE *js = argc > 1 ? fopen(argv[1], "w") : nullptr;
    if (js) fprintf(js, "{\"device\": \"%s\", \"cus\": %d, \"classes\": {", prop.name, cus);
    bool first = true;
    for (auto &e : es) {
        printf("%-30s", e.name);
        double rate8 = 0, mhz = 0, r[4] = {0, 0, 0, 0};
        int wi = 0;
        for (int wps : {1, 2, 4, 8}) {
            const int blocks = cus * wps;
            hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, 10, out, clk);   // warm
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, iters, out, clk);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(&hclk, clk, 8, hipMemcpyDeviceToHost);
            // the kernel's own clock count for one wave ~ the kernel duration (all waves run the same loop)
            const double insts_per_simd = (double) iters * 64 * e.per_stmt * wps;
            const double rate = insts_per_simd / (double) hclk;
            mhz = (double) hclk / (ms * 1e3);
            rate8 = rate;
            r[wi++] = rate;
            printf(" %8.3f", rate);
        }
        const double tlane = rate8 * 64 * 4 * cus * mhz * 1e6 / 1e12;
        printf("   %.0f / %.1f\n", mhz, tlane);
        fflush(stdout);
        if (js) fprintf(js, "%s\"%s\": {\"wave_insts_per_clk_per_simd\": [%.4f, %.4f, %.4f, %.4f], \"clk_mhz\": %.0f, \"chip_Tlaneops_s\": %.2f}",
                        first ? "" : ", ", e.name, r[0], r[1], r[2], r[3], mhz, tlane);
        first = false;
    }
    if (js) { fprintf(js, "}}\n"); fclose(js); }
    return 0;
}
