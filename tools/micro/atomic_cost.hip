// micro-benchmark: what a returning atomicAdd on a global work counter costs a persistent grid.
//   hipcc --offload-arch=gfx950 -O3 -o atomic_cost atomic_cost.hip && ./atomic_cost
// One counter: device-scope RMWs execute at the memory side and serialise (~11 ns each however many CUs ask).  k_slice's tile
// dealer uses 64 counters: this measures 64 counters at different strides (do counters that share a line / a channel
// serialise with each other?) and, for reference, the same with workgroup scope (the RMW then runs in the XCD's L2; only
// meaningful as a speed limit: the XCDs' L2s are not coherent with each other).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int SCOPE>
__global__ void k(unsigned *ctr, unsigned ncounters, unsigned stride_words, unsigned total_per_counter, unsigned long long *sink, int spin) {
    const int lane = threadIdx.x & 63;
    unsigned *my = ctr + (size_t) (blockIdx.x % ncounters) * stride_words;
    unsigned long long acc = 0;
    for (;;) {
        unsigned u = 0;
        if (lane == 0) u = __hip_atomic_fetch_add(my, 1u, __ATOMIC_RELAXED, SCOPE);
        u = __builtin_amdgcn_readfirstlane(u);
        if (u >= total_per_counter) break;
        for (int i = 0; i < spin; ++i) acc += __builtin_amdgcn_s_memtime();   // stand-in for a unit's work
    }
    if (lane == 0 && acc == 12345) *sink = acc;
}
int main() {
    unsigned *ctr; unsigned long long *sink;
    const size_t bytes = 64 * 4096 + 4096;
    (void) hipMalloc(&ctr, bytes); (void) hipMalloc(&sink, 8);
    hipEvent_t a, b; (void) hipEventCreate(&a); (void) hipEventCreate(&b);
    const unsigned total = 262144;
    for (int scope = 0; scope < 2; ++scope)
    for (int spin : {0, 2000}) for (unsigned nc : {1u, 8u, 64u}) for (unsigned stride : {1u, 16u, 64u, 1024u}) {
        if (nc == 1 && stride != 1) continue;
        (void) hipMemset(ctr, 0, bytes);
        (void) hipEventRecord(a);
        if (scope == 0) hipLaunchKernelGGL(k<__HIP_MEMORY_SCOPE_AGENT>, dim3(768), dim3(256), 0, 0, ctr, nc, stride, total / nc, sink, spin);
        else hipLaunchKernelGGL(k<__HIP_MEMORY_SCOPE_WORKGROUP>, dim3(768), dim3(256), 0, 0, ctr, nc, stride, total / nc, sink, spin);
        (void) hipEventRecord(b); (void) hipEventSynchronize(b);
        float ms; (void) hipEventElapsedTime(&ms, a, b);
        printf("%s scope, spin %5d, %2u counters, stride %5u B: %8.3f ms for %u grabs (%.2f ns per grab, all counters together)\n",
               scope == 0 ? "agent    " : "workgroup", spin, nc, stride * 4, ms, total, ms * 1e6 / total);
    }
    return 0;
}
