// tools/micro/valu_issue.hip — what one instruction costs a SIMD on gfx950, by wall clock.
//   hipcc --offload-arch=gfx950 -O3 -o valu_issue valu_issue.hip && ./valu_issue [out.json]
// For each instruction mix: a loop body of 64 statements (8 independent accumulators x 8) run by 1, 2, 4, 6 and 8 waves per
// SIMD on every CU, timed with HIP events (the kernels' own s_memtime turned out not to be the shader clock on this part:
// round 2's "per clock" columns were meaningless, only the wall-clock column stood).  Reported per mix and occupancy:
// chip-wide T lane-ops/s of the VALU instructions in the mix, and ns per loop statement per SIMD — the second is what the
// sweep / slice kernels are budgeted with (their time is close to the sum of ALL their instructions, scalar ones included,
// at one price: see DESIGN.md §3).  The mixes with scalar / nop / waitcnt filler say what those cost beside VALU work.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>

// one asm statement per accumulator; a[i] is read and written, b and c are loop-invariant inputs
#define DEF_KERNEL(NAME, ASM)                                                                       \
    __global__ __launch_bounds__(256) void NAME(int iters, uint32_t *out) {                          \
        uint32_t a[8], b = threadIdx.x * 2654435761u + 12345u, c = blockIdx.x * 40503u + 977u;       \
        for (int i = 0; i < 8; ++i) a[i] = b ^ (i * 0x9e3779b9u);                                     \
        for (int it = 0; it < iters; ++it) {                                                          \
            _Pragma("unroll") for (int r = 0; r < 8; ++r) {                                           \
                _Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile(ASM : "+v"(a[i]) : "v"(b), "v"(c) : "vcc", "scc", "s20", "s21", "s22", "s23"); \
            }                                                                                         \
        }                                                                                             \
        uint32_t x = 0;                                                                               \
        for (int i = 0; i < 8; ++i) x ^= a[i];                                                        \
        if (x == 0x12345678u) out[threadIdx.x] = x;                                                   \
    }

// the same loop with ONE accumulator: every statement depends on the one before it
#define DEF_CHAIN(NAME, ASM)                                                                        \
    __global__ __launch_bounds__(256) void NAME(int iters, uint32_t *out) {                          \
        uint32_t a = threadIdx.x * 2654435761u + 12345u, b = a ^ 0x9e3779b9u, c = blockIdx.x * 40503u + 977u; \
        for (int it = 0; it < iters; ++it) {                                                          \
            _Pragma("unroll") for (int r = 0; r < 64; ++r) asm volatile(ASM : "+v"(a) : "v"(b), "v"(c) : "vcc", "scc", "s20", "s21", "s22", "s23"); \
        }                                                                                             \
        if (a == 0x12345678u) out[threadIdx.x] = a;                                                   \
    }
DEF_CHAIN(c_add_u32, "v_add_u32 %0, %0, %1")
DEF_CHAIN(c_pk_max, "v_pk_max_i16 %0, %0, %1")
DEF_CHAIN(c_alignbit, "v_alignbit_b32 %0, %0, %1, %2")
DEF_CHAIN(c_dot2c, "v_dot2c_i32_i16 %0, %1, %2")
DEF_CHAIN(c_dot2, "v_dot2_i32_i16 %0, %1, %2, %0")
DEF_CHAIN(c_dot2c_then_valu, "v_dot2c_i32_i16 %0, %1, %2\n s_nop 2\n v_xor_b32 %0, %0, %1")

DEF_KERNEL(k_add_u32, "v_add_u32 %0, %0, %1")
DEF_KERNEL(k_xor, "v_xor_b32 %0, %0, %1")
DEF_KERNEL(k_xor_lit, "v_xor_b32 %0, 0x80008000, %0")
DEF_KERNEL(k_and_or, "v_and_or_b32 %0, %0, %1, %2")
DEF_KERNEL(k_lshl_or, "v_lshl_or_b32 %0, %0, 5, %1")
DEF_KERNEL(k_alignbit, "v_alignbit_b32 %0, %0, %1, %2")
DEF_KERNEL(k_perm, "v_perm_b32 %0, %0, %1, %2")
DEF_KERNEL(k_bfe, "v_bfe_u32 %0, %0, 3, 16")
DEF_KERNEL(k_pk_sub_i16, "v_pk_sub_i16 %0, %0, %1 clamp")
DEF_KERNEL(k_pk_max_i16, "v_pk_max_i16 %0, %0, %1")
DEF_KERNEL(k_pk_ashr, "v_pk_ashrrev_i16 %0, 15, %0 op_sel_hi:[0,1]")
DEF_KERNEL(k_dot2_i16, "v_dot2_i32_i16 %0, %1, %2, %0")
DEF_KERNEL(k_dot2c, "v_dot2c_i32_i16 %0, %1, %2")
DEF_KERNEL(k_dot2c_lit, "v_dot2c_i32_i16 %0, 0x200020, %1")
DEF_KERNEL(k_ffbl, "v_ffbl_b32 %0, %0")
DEF_KERNEL(k_bcnt, "v_bcnt_u32_b32 %0, %1, %0")
DEF_KERNEL(k_mul_lo, "v_mul_lo_u32 %0, %0, %1")
DEF_KERNEL(k_mad_u24, "v_mad_u32_u24 %0, %0, %1, %2")
DEF_KERNEL(k_mbcnt, "v_mbcnt_lo_u32_b32 %0, %1, %0")
DEF_KERNEL(k_cmp_cndmask, "v_cmp_gt_i32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc")
DEF_KERNEL(k_dpp_add, "v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1")
DEF_KERNEL(k_readlane, "v_readlane_b32 s20, %0, 3\n v_add_u32 %0, s20, %0")
// the same VALU statement with one non-VALU instruction beside it: what scalar work, nops and (idle) waits cost a busy SIMD
DEF_KERNEL(k_add_salu, "v_add_u32 %0, %0, %1\n s_add_u32 s20, s20, 1")
DEF_KERNEL(k_add_salu2, "v_add_u32 %0, %0, %1\n s_add_u32 s20, s20, 1\n s_and_b32 s21, s21, s20")
DEF_KERNEL(k_add_nop, "v_add_u32 %0, %0, %1\n s_nop 0")
DEF_KERNEL(k_add_nop1, "v_add_u32 %0, %0, %1\n s_nop 1")
DEF_KERNEL(k_add_wait, "v_add_u32 %0, %0, %1\n s_waitcnt lgkmcnt(0)")
DEF_KERNEL(k_pk_salu, "v_pk_max_i16 %0, %0, %1\n s_add_u32 s20, s20, 1")
DEF_KERNEL(k_pk_nop, "v_pk_max_i16 %0, %0, %1\n s_nop 0")
DEF_KERNEL(k_pk_wait, "v_pk_max_i16 %0, %0, %1\n s_waitcnt lgkmcnt(0)")
DEF_KERNEL(k_salu_only, "s_add_u32 s20, s20, 1")

typedef void (*kern_t)(int, uint32_t *);
struct Entry { const char *name; kern_t k; int valu_per_stmt; };

int main(int argc, char **argv) {
    Entry es[] = {
        {"v_add_u32 (VOP2)", k_add_u32, 1}, {"v_xor_b32 (VOP2)", k_xor, 1}, {"v_xor_b32 literal (VOP2 + 4 B)", k_xor_lit, 1},
        {"v_and_or_b32 (VOP3)", k_and_or, 1}, {"v_lshl_or_b32 (VOP3)", k_lshl_or, 1}, {"v_alignbit_b32 (VOP3)", k_alignbit, 1},
        {"v_perm_b32 (VOP3)", k_perm, 1}, {"v_bfe_u32 (VOP3)", k_bfe, 1}, {"v_pk_sub_i16 clamp (VOP3P)", k_pk_sub_i16, 1},
        {"v_pk_max_i16 (VOP3P)", k_pk_max_i16, 1}, {"v_pk_ashrrev_i16 (VOP3P)", k_pk_ashr, 1}, {"v_dot2_i32_i16 (VOP3P)", k_dot2_i16, 1},
        {"v_dot2c_i32_i16 (VOP2)", k_dot2c, 1}, {"v_dot2c_i32_i16 literal (VOP2 + 4 B)", k_dot2c_lit, 1}, {"v_ffbl_b32 (VOP1)", k_ffbl, 1},
        {"v_bcnt_u32_b32 (VOP3)", k_bcnt, 1}, {"v_mul_lo_u32 (VOP3)", k_mul_lo, 1}, {"v_mad_u32_u24 (VOP3)", k_mad_u24, 1},
        {"v_mbcnt_lo (VOP3)", k_mbcnt, 1}, {"v_cmp + v_cndmask (vcc)", k_cmp_cndmask, 2}, {"v_add_u32 dpp row_shr (independent)", k_dpp_add, 1},
        {"v_readlane + v_add", k_readlane, 2},
        {"v_add_u32 + s_add_u32", k_add_salu, 1}, {"v_add_u32 + 2 SALU", k_add_salu2, 1}, {"v_add_u32 + s_nop 0", k_add_nop, 1},
        {"v_add_u32 + s_nop 1", k_add_nop1, 1}, {"v_add_u32 + s_waitcnt (idle)", k_add_wait, 1},
        {"v_pk_max_i16 + s_add_u32", k_pk_salu, 1}, {"v_pk_max_i16 + s_nop 0", k_pk_nop, 1}, {"v_pk_max_i16 + s_waitcnt (idle)", k_pk_wait, 1},
        {"s_add_u32 alone", k_salu_only, 0},
        {"chain: v_add_u32", c_add_u32, 1}, {"chain: v_pk_max_i16", c_pk_max, 1}, {"chain: v_alignbit_b32", c_alignbit, 1},
        {"chain: v_dot2c_i32_i16 (accumulate)", c_dot2c, 1}, {"chain: v_dot2_i32_i16 (VOP3P, acc = src2)", c_dot2, 1},
        {"chain: v_dot2c, s_nop 2, v_xor", c_dot2c_then_valu, 2},
    };
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    uint32_t *out;
    hipMalloc(&out, 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    const int occ[5] = {1, 2, 4, 6, 8};
    printf("# %s, %d CUs; loop body = 64 statements, %d iterations; timed with HIP events\n", prop.name, cus, iters);
    printf("# per occupancy (waves per SIMD): chip T lane-ops/s of the VALU instructions | ns per statement per SIMD\n");
    printf("%-40s %14s %14s %14s %14s %14s\n", "mix", "1", "2", "4", "6", "8");
    FILE *js = argc > 1 ? fopen(argv[1], "w") : nullptr;
    if (js) fprintf(js, "{\"device\": \"%s\", \"cus\": %d, \"waves_per_simd\": [1, 2, 4, 6, 8], \"mixes\": {", prop.name, cus);
    bool first = true;
    for (auto &e : es) {
        printf("%-40s", e.name);
        double tl[5], ns[5];
        for (int wi = 0; wi < 5; ++wi) {
            const int wps = occ[wi];
            const int blocks = cus * wps;                      // 256 threads = 4 waves = one per SIMD
            hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, 10, out);   // warm
            hipDeviceSynchronize();
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, iters, out);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            const double stmts_per_simd = (double) iters * 64 * wps;
            ns[wi] = best * 1e6 / stmts_per_simd;
            tl[wi] = stmts_per_simd * e.valu_per_stmt * 64.0 * 4 * cus / (best * 1e-3) / 1e12;
            printf(" %6.1f | %5.2f", tl[wi], ns[wi]);
        }
        printf("\n");
        fflush(stdout);
        if (js) fprintf(js, "%s\"%s\": {\"chip_Tlaneops_s\": [%.2f, %.2f, %.2f, %.2f, %.2f], \"ns_per_statement_per_simd\": [%.3f, %.3f, %.3f, %.3f, %.3f]}",
                        first ? "" : ", ", e.name, tl[0], tl[1], tl[2], tl[3], tl[4], ns[0], ns[1], ns[2], ns[3], ns[4]);
        first = false;
    }
    if (js) { fprintf(js, "}}\n"); fclose(js); }
    return 0;
}
