// micro-benchmark: cost of returning device-scope atomicAdd on ONE address from a persistent grid
// (is a global work counter affordable?).  hipcc --offload-arch=gfx950 -O3 -o atomic_cost atomic_cost.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned *ctr, unsigned total, unsigned long long *sink, int spin) {
    const int lane = threadIdx.x & 63;
    unsigned long long acc = 0;
    for (;;) {
        unsigned u = 0;
        if (lane == 0) u = atomicAdd(ctr, 1u);
        u = __builtin_amdgcn_readfirstlane(u);
        if (u >= total) break;
        for (int i = 0; i < spin; ++i) acc += __builtin_amdgcn_s_memtime();   // stand-in for a unit's work
    }
    if (lane == 0 && acc == 12345) *sink = acc;
}
int main() {
    unsigned *ctr; unsigned long long *sink;
    hipMalloc(&ctr, 4); hipMalloc(&sink, 8);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int spin : {0, 2000, 20000}) for (unsigned total : {8192u, 65536u, 1000000u}) {
        hipMemset(ctr, 0, 4);
        hipEventRecord(a);
        hipLaunchKernelGGL(k, dim3(768), dim3(256), 0, 0, ctr, total, sink, spin);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("spin %6d  grabs %8u : %.3f ms  (%.1f ns per grab)\n", spin, total, ms, ms * 1e6 / total);
    }
    return 0;
}
