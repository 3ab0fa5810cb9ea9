// tools/micro/copy_bw.hip — what a plain copy reaches on this GPU with k_sweep_uc8's traffic: 537 MB read + 537 MB written per launch
// (16 bytes per lane and access), the launches walking three source and three destination arrays round-robin like
// tools/micro/sweep_uc8_cold.hip.  The calibration for "the fused sweep sits on its memory time": the kernel's 1:1 read / write
// mix cannot go faster than this.
//   hipcc --offload-arch=gfx950 -O3 -o copy_bw copy_bw.hip ;  copy_bw [MB per array = 512] [launches = 300]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// grid-stride: every workgroup walks the array in strides of the whole grid (one contiguous window of the array at any time)
__global__ __launch_bounds__(256) void k_copy_stride(const u32x4 *src, u32x4 *dst, size_t n16) {
    for (size_t i = (size_t) blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t) gridDim.x * 256) dst[i] = src[i];
}
// ... the same with four loads in flight per lane
__global__ __launch_bounds__(256) void k_copy_stride4(const u32x4 *src, u32x4 *dst, size_t n16) {
    const size_t step = (size_t) gridDim.x * 256;
    size_t i = (size_t) blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * step < n16; i += 4 * step) {
        const u32x4 a = src[i], b = src[i + step], c = src[i + 2 * step], d = src[i + 3 * step];
        dst[i] = a; dst[i + step] = b; dst[i + 2 * step] = c; dst[i + 3 * step] = d;
    }
    for (; i < n16; i += step) dst[i] = src[i];
}
// one element per thread (a grid as large as the array)
__global__ __launch_bounds__(256) void k_copy_flat(const u32x4 *src, u32x4 *dst, size_t n16) {
    const size_t i = (size_t) blockIdx.x * 256 + threadIdx.x;
    if (i < n16) dst[i] = src[i];
}

// non-persistent, but K accesses per thread: workgroup b copies the contiguous piece [b, b + 1) x K x BLOCK x 16 bytes, one BLOCK x 16 B row after the other
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_copy_chunk(const u32x4 *src, u32x4 *dst, size_t n16, int K) {
    size_t i = (size_t) blockIdx.x * BLOCK * K + threadIdx.x;
    for (int k = 0; k < K && i < n16; ++k, i += BLOCK) dst[i] = src[i];
}
// ... every WAVE its own contiguous piece of K KB (the fused sweep's shape: a wave walks contiguous steps)
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_copy_wavechunk(const u32x4 *src, u32x4 *dst, size_t n16, int K) {
    const size_t wave = (size_t) blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6);
    size_t i = wave * 64 * K + (threadIdx.x & 63);
    for (int k = 0; k < K && i < n16; ++k, i += 64) dst[i] = src[i];
}
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_copy_flat_b(const u32x4 *src, u32x4 *dst, size_t n16) {
    const size_t i = (size_t) blockIdx.x * BLOCK + threadIdx.x;
    if (i < n16) dst[i] = src[i];
}

int main(int argc, char **argv) {
    const size_t mb = argc > 1 ? (size_t) atoi(argv[1]) : 512;
    const int launches = argc > 2 ? atoi(argv[2]) : 300;
    const size_t bytes = mb << 20, n16 = bytes / 16;
    const int R = 3;
    char *s, *d;
    CK(hipMalloc(&s, R * bytes));
    CK(hipMalloc(&d, R * bytes));
    CK(hipMemset(s, 1, R * bytes));
    CK(hipMemset(d, 2, R * bytes));
    std::vector<hipEvent_t> ev(2 * (size_t) launches);
    for (auto &e : ev) CK(hipEventCreate(&e));
    struct V { const char *name; int kind; unsigned grid; } vs[] = {
        {"grid-stride, 256 CUs x 8 workgroups", 0, 2048}, {"grid-stride, 256 CUs x 4 workgroups", 0, 1024}, {"grid-stride, 256 CUs x 16 workgroups", 0, 4096},
        {"grid-stride x4 in flight, 256 CUs x 4 workgroups", 1, 1024}, {"grid-stride x4 in flight, 256 CUs x 8 workgroups", 1, 2048},
        {"one element per thread", 2, (unsigned) ((n16 + 255) / 256)},
        {"one element per thread, workgroups of 1024", 3, (unsigned) ((n16 + 1023) / 1024)},
        {"workgroup of 256 = a contiguous piece, 4 rows", 10, 4}, {"workgroup of 256 = a contiguous piece, 16 rows", 10, 16}, {"workgroup of 256 = a contiguous piece, 64 rows", 10, 64},
        {"workgroup of 1024 = a contiguous piece, 4 rows", 11, 4}, {"workgroup of 1024 = a contiguous piece, 16 rows", 11, 16}, {"workgroup of 1024 = a contiguous piece, 64 rows", 11, 64},
        {"wave = a contiguous piece of 8 KB, workgroups of 1024", 12, 8}, {"wave = a contiguous piece of 32 KB, workgroups of 1024", 12, 32}, {"wave = a contiguous piece of 128 KB, workgroups of 1024", 12, 128},
        {"wave = a contiguous piece of 32 KB, workgroups of 256", 13, 32},
        {"grid-stride, 256 workgroups of 1024", 4, 256}};
    printf("{\"bytes_read_per_launch\": %zu, \"bytes_written_per_launch\": %zu, \"arrays\": %d, \"launches\": %d, \"variants\": [", bytes, bytes, R, launches);
    bool first = true;
    for (const V &v : vs) {
        for (int pass = 0; pass < 2; ++pass) {                // (pass 0: warm-up of the clocks)
            for (int k = 0; k < launches; ++k) {
                const u32x4 *sp = (const u32x4 *) (s + (size_t) (k % R) * bytes);
                u32x4 *dp = (u32x4 *) (d + (size_t) (k % R) * bytes);
                CK(hipEventRecord(ev[2 * k], nullptr));
                if (v.kind == 0) hipLaunchKernelGGL(k_copy_stride, dim3(v.grid), dim3(256), 0, nullptr, sp, dp, n16);
                else if (v.kind == 1) hipLaunchKernelGGL(k_copy_stride4, dim3(v.grid), dim3(256), 0, nullptr, sp, dp, n16);
                else if (v.kind == 2) hipLaunchKernelGGL(k_copy_flat, dim3(v.grid), dim3(256), 0, nullptr, sp, dp, n16);
                else if (v.kind == 3) hipLaunchKernelGGL(k_copy_flat_b<1024>, dim3(v.grid), dim3(1024), 0, nullptr, sp, dp, n16);
                else if (v.kind == 4) hipLaunchKernelGGL(k_copy_stride, dim3(v.grid * 4), dim3(256), 0, nullptr, sp, dp, n16);
                else if (v.kind == 10) hipLaunchKernelGGL(k_copy_chunk<256>, dim3((unsigned) ((n16 + 256 * v.grid - 1) / (256 * v.grid))), dim3(256), 0, nullptr, sp, dp, n16, (int) v.grid);
                else if (v.kind == 11) hipLaunchKernelGGL(k_copy_chunk<1024>, dim3((unsigned) ((n16 + 1024 * v.grid - 1) / (1024 * v.grid))), dim3(1024), 0, nullptr, sp, dp, n16, (int) v.grid);
                else if (v.kind == 12) hipLaunchKernelGGL(k_copy_wavechunk<1024>, dim3((unsigned) ((n16 + 1024 * v.grid - 1) / (1024 * v.grid))), dim3(1024), 0, nullptr, sp, dp, n16, (int) v.grid);
                else if (v.kind == 13) hipLaunchKernelGGL(k_copy_wavechunk<256>, dim3((unsigned) ((n16 + 256 * v.grid - 1) / (256 * v.grid))), dim3(256), 0, nullptr, sp, dp, n16, (int) v.grid);
                CK(hipEventRecord(ev[2 * k + 1], nullptr));
            }
            CK(hipDeviceSynchronize());
        }
        std::vector<float> us;
        for (int k = 0; k < launches; ++k) { float ms; CK(hipEventElapsedTime(&ms, ev[2 * k], ev[2 * k + 1])); us.push_back(ms * 1e3f); }
        std::sort(us.begin(), us.end());
        double avg = 0;
        for (float x : us) avg += x;
        avg /= us.size();
        printf("%s{\"copy\": \"%s\", \"us_mean\": %.1f, \"us_median\": %.1f, \"us_min\": %.1f, \"TBs_read_plus_written\": %.3f, \"frac_of_8TBs\": %.3f}", first ? "" : ", ", v.name, avg, us[us.size() / 2], us.front(),
               2.0 * bytes / (avg * 1e-6) / 1e12, 2.0 * bytes / (avg * 1e-6) / 1e12 / 8.0);
        first = false;
    }
    printf("]}\n");
    return 0;
}
