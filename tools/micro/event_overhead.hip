// tools/micro/event_overhead.hip — what a pair of timing HIP events around ONE kernel in a busy stream reports beyond the kernel's own
// duration.  The kernel spins for a known time on the 100 MHz s_memrealtime counter; the stream holds a kernel before and behind
// the bracket, as the library's main stream does (k_convert | ev | k_sweep | ev | k_slice).
//   hipcc --offload-arch=gfx950 -O3 -o event_overhead event_overhead.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

__global__ void spin(unsigned ticks, unsigned long long *sink) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) { }
    if (sink && threadIdx.x == 0 && blockIdx.x == 0) *sink = t0;
}

int main() {
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    unsigned long long *sink;
    CK(hipMalloc(&sink, 8));
    const unsigned blocks = 1536;
    for (int r = 0; r < 20; ++r) hipLaunchKernelGGL(spin, dim3(blocks), dim3(256), 0, s, 1000u, sink);
    CK(hipStreamSynchronize(s));
    for (unsigned us : {0u, 10u, 40u, 100u}) {
        std::vector<float> v;
        for (int r = 0; r < 200; ++r) {
            hipLaunchKernelGGL(spin, dim3(blocks), dim3(256), 0, s, 5000u, sink);           // 50 us of something before
            CK(hipEventRecord(a, s));
            if (us) hipLaunchKernelGGL(spin, dim3(blocks), dim3(256), 0, s, us * 100u, sink);
            CK(hipEventRecord(b, s));
            hipLaunchKernelGGL(spin, dim3(blocks), dim3(256), 0, s, 5000u, sink);           // and behind
            CK(hipStreamSynchronize(s));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, a, b));
            v.push_back(ms * 1e3f);
        }
        std::sort(v.begin(), v.end());
        printf("kernel spins %3u us: events report median %.2f us (min %.2f, p90 %.2f)\n", us, v[v.size() / 2], v[0], v[v.size() * 9 / 10]);
    }
    return 0;
}
