// tools/micro/cu_mask_map.hip — where do the bits of a hipExtStreamCreateWithCUMask mask land?  For a few masks (every k-th bit from
// bit off), a kernel of many short workgroups on the masked stream records every workgroup's XCC_ID and HW_ID (shader engine, CU):
// the table says which XCDs / CUs a mask really selects — the pipeline's side streams use "every 8th CU" and "every 4th CU from 2".
//   hipcc --offload-arch=gfx950 -O3 -o cu_mask_map cu_mask_map.hip ; cu_mask_map
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <map>
#include <set>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

__global__ void k_where(unsigned *out) {
    if (threadIdx.x == 0) {
        unsigned xcc, hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hw;
    }
    // stay a little, so that the launch spreads over everything the mask allows
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < 2000) { }
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const unsigned nb = 8192;
    unsigned *d;
    CK(hipMalloc(&d, nb * 8));
    std::vector<unsigned> h(2 * nb);
    struct M { int k, off; } masks[] = {{1, 0}, {8, 0}, {8, 4}, {4, 2}, {2, 1}, {3, 1}, {16, 0}, {64, 0}, {0, 0} /* bits 0..31 */};
    printf("CUs %d\n", cus);
    for (const M &m : masks) {
        uint32_t mask[32] = {0};
        int nbits = 0;
        if (m.k == 0) { mask[0] = 0xffffffffu; nbits = 32; }
        else for (int cu = m.off; cu < cus; cu += m.k) { mask[cu >> 5] |= 1u << (cu & 31); ++nbits; }
        hipStream_t s;
        CK(hipExtStreamCreateWithCUMask(&s, (uint32_t) ((cus + 31) / 32), mask));
        CK(hipMemsetAsync(d, 0xff, nb * 8, s));
        hipLaunchKernelGGL(k_where, dim3(nb), dim3(64), 0, s, d);
        CK(hipStreamSynchronize(s));
        CK(hipMemcpy(h.data(), d, nb * 8, hipMemcpyDeviceToHost));
        std::map<unsigned, std::set<unsigned>> per_xcc;      // XCC -> distinct (SE, SH, CU)
        for (unsigned b = 0; b < nb; ++b) {
            const unsigned xcc = h[2 * b] & 0xf, hw = h[2 * b + 1];
            const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
            per_xcc[xcc].insert(se << 8 | sh << 4 | cu);
        }
        if (m.k) printf("mask every %d-th bit from %d (%d bits):", m.k, m.off, nbits); else printf("mask bits 0..31:");
        unsigned total = 0;
        for (auto &kv : per_xcc) { printf("  XCC %u: %zu CUs", kv.first, kv.second.size()); total += (unsigned) kv.second.size(); }
        printf("  | %u distinct CUs ran workgroups\n", total);
        CK(hipStreamDestroy(s));
    }
    return 0;
}
