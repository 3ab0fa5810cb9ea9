// tools/micro/sweep_uc8_cold.hip — k_sweep_uc8 (converter and sweep in one pass: the kernel bench.py's roofline is quoted on)
// on its own, over COLD memory, with nothing beside it.
//
// Why: in the pipeline the kernel runs beside the side streams' small kernels (on hardware queues of their own, which has no effect
// under rocprofv3) — under the profiler the pipeline's k_sweep_uc8 launches average 285 us where the benchmark's events
// read 207-212 (profiles/README.md).  Here the kernel is the only thing on the GPU, so `rocprofv3 --kernel-trace --stats` of THIS
// program and the HIP events around each launch measure the same thing and have to agree.
// A launch reads 2 B and writes 2 B per sample: at 2048 buffers 537 MB + 537 MB, four times the 256 MiB Infinity Cache by itself;
// the launches also walk R replicas of the IQ block and R magnitude arrays round-robin, so every launch touches memory last used
// R - 1 launches (> 2 GB of traffic) earlier.
// Every replica's first launch is checked against the CPU: all magnitudes against init_uc8_lookup's table (convert.c:35-62), the
// per-buffer sum(mag) / sum(mag^2) (convert.c:64-108), and the candidate lists against demodulate2400's pre-check + threshold tests
// (demod_2400.c:311-378, restated in check_cpu).
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DMGPU_EXPERIMENTS=1 -o sweep_uc8_cold sweep_uc8_cold.hip -ldl
//   sweep_uc8_cold [buffers=2048] [replicas=3] [rounds=5] [dense=0] [rate=2000] [queued=1]
// Output: one JSON line (per-launch time, GB/s of algorithmic bytes = 4 B per sample, fraction of the 8 TB/s peak).
#include "../../readsb_amd/csrc/kernels.hip"
#include "../../readsb_amd/csrc/tables.cpp"

#include <dlfcn.h>
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

using namespace mgpu;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

typedef int (*synth_fn)(uint64_t, int, double, int, int, double, uint64_t, uint64_t, void *, int);

static void check_cpu(const uint16_t *m, uint64_t n, int thr, std::vector<uint32_t> &out) {
    out.clear();
    for (uint64_t D = 0; D < n; ++D) {
        const uint16_t *pa = m + D;
        if (!(pa[1] > pa[7] && pa[12] > pa[14] && pa[12] > pa[15])) continue;
        const int32_t base_noise = pa[5] + pa[8] + pa[16] + pa[17] + pa[18];
        const int32_t ref = (base_noise * thr) >> 5;
        const int32_t diff_2_3 = pa[2] - pa[3], sum_1_4 = pa[1] + pa[4], diff_10_11 = pa[10] - pa[11];
        const int32_t common = sum_1_4 - diff_2_3 + pa[9] + pa[12];
        uint32_t mask = 0;
        if (common - diff_10_11 >= ref) mask |= 1;
        if (common + diff_10_11 >= ref) mask |= 2;
        if (sum_1_4 + 2 * diff_2_3 + diff_10_11 + pa[12] >= ref) mask |= 4;
        if (mask) out.push_back((uint32_t) (D << 3) | mask);
    }
}

int main(int argc, char **argv) {
    const int buffers = argc > 1 ? atoi(argv[1]) : 2048;
    const int replicas = argc > 2 ? atoi(argv[2]) : 3;
    const int rounds = argc > 3 ? atoi(argv[3]) : 5;
    const int dense = argc > 4 ? atoi(argv[4]) : 0;
    const double rate = argc > 5 ? atof(argv[5]) : 2000.0;
    const int queued = argc > 6 ? atoi(argv[6]) : 1;         // 1: the timed rounds' launches are enqueued back to back (the GPU never idles between them, as in the pipeline); 0: one at a time with a host wait behind each
    const int thr = 58;
    const int iq_shift = getenv("UC8_IQ_SHIFT") ? atoi(getenv("UC8_IQ_SHIFT")) : 0;   // timing experiment: bytes added to the IQ pointer (12: the kernel's 16-byte loads become 16-byte aligned; the results are those of a shifted stream: mismatches reported)
    const uint32_t buf_samples = 131072;
    const uint64_t n = (uint64_t) buffers * buf_samples;
    const uint64_t stride = ((n + kTrailing + 4096 + 4095) / 4096) * 4096;      // magnitudes per replica (16-byte aligned, with the tile slack)

    std::string here = argv[0];
    here = here.substr(0, here.find_last_of('/') == std::string::npos ? 0 : here.find_last_of('/'));
    const std::string so = (here.empty() ? std::string(".") : here) + "/../libsynth_iq.so";
    void *h = dlopen(so.c_str(), RTLD_NOW);
    if (!h) { fprintf(stderr, "%s: %s\n", so.c_str(), dlerror()); return 2; }
    synth_fn synth = (synth_fn) dlsym(h, "synth_iq_generate");
    if (!synth) { fprintf(stderr, "synth_iq_generate: %s\n", dlerror()); return 2; }
    std::vector<uint8_t> iq(n * 2);
    synth(424242, 0, rate, 200, dense, 3.0, 0, n, iq.data(), 32);

    // ---- what the CPU says: magnitudes, per-buffer sums, candidates ----
    const uint16_t *tab = uc8_table();
    std::vector<uint16_t> want_mag(stride, 0);                                  // [0, 326): the samples before the stream (zeros, sdr_ifile.c:209-213)
    std::vector<unsigned long long> want_level(buffers, 0), want_power(buffers, 0);
    for (uint64_t i = 0; i < n; ++i) {
        const uint16_t m = tab[(uint32_t) iq[2 * i] | ((uint32_t) iq[2 * i + 1] << 8)];
        want_mag[kTrailing + i] = m;
        want_level[i / buf_samples] += m;
        want_power[i / buf_samples] += (unsigned long long) m * m;
    }
    std::vector<uint32_t> want;
    check_cpu(want_mag.data(), n, thr, want);

    uint8_t *d_iq;
    uint16_t *d_mag, *d_cand, *d_lut;
    uint32_t *d_count, *d_part, *d_dealer;
    unsigned long long *d_sums, *d_waves;
    const uint32_t nsteps = (uint32_t) ((n + kSwStep - 1) / kSwStep);
    CK(hipMalloc(&d_iq, (size_t) replicas * n * 2 + 4096));                        // (slack: UC8_IQ_SHIFT)
    CK(hipMalloc(&d_mag, (size_t) replicas * stride * 2));
    CK(hipMalloc(&d_cand, (size_t) (nsteps + 2) * kSwStep * 2));
    CK(hipMalloc(&d_count, (size_t) (nsteps + 2) * 4));
    CK(hipMalloc(&d_part, 65536 * 8 * 4));
    CK(hipMalloc(&d_dealer, (size_t) 2 * kDealerCounters * kDealerStride * 4));
    CK(hipMalloc(&d_sums, (size_t) (buffers + 3) * 2 * 8));
    CK(hipMalloc(&d_waves, 65536 * 4 * 2 * 8));
    const std::vector<uint16_t> lut = uc8_folded_table();
    CK(hipMalloc(&d_lut, lut.size() * 2));
    CK(hipMemcpy(d_lut, lut.data(), lut.size() * 2, hipMemcpyHostToDevice));
    for (int r = 0; r < replicas; ++r) CK(hipMemcpy(d_iq + (size_t) r * n * 2, iq.data(), n * 2, hipMemcpyHostToDevice));
    CK(hipMemset(d_mag, 0xA5, (size_t) replicas * stride * 2));                 // (the kernel writes the tail too: nothing of this may survive)

    SweepParams p{};
    p.n = n; p.thr = thr; p.cand = d_cand; p.cand_count = d_count; p.sweep_part = d_part; p.dealer = d_dealer;
    p.dbg_waves = nullptr;
    p.iq_format = 0; p.tail = nullptr; p.uc8_sym = d_lut + UC8_SYM_OFFSET;
    p.sum_level = d_sums; p.sum_power = d_sums + buffers + 3;
    p.buf_steps = buf_samples / (uint32_t) kSweepTile;
    (void) d_waves;

    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    unsigned blocks = 0;
    auto run = [&](int r, float &us) -> int {
        p.iq = d_iq + (size_t) r * n * 2 + iq_shift;
        p.mag = d_mag + (size_t) r * stride;
        p.mag_w = d_mag + (size_t) r * stride;
        CK(hipMemsetAsync(d_dealer, 0, (size_t) 2 * kDealerCounters * kDealerStride * 4, nullptr));   // (the pipeline's k_publish hands both back zeroed)
        CK(hipMemsetAsync(d_sums, 0, (size_t) (buffers + 3) * 2 * 8, nullptr));
        CK(hipEventRecord(e0, nullptr));
        blocks = launch_sweep(p, nullptr);                   // the library's launcher: resident grid, pacing from its running estimate
        CK(hipEventRecord(e1, nullptr));
        CK(hipEventSynchronize(e1));
        CK(hipGetLastError());
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        us = ms * 1e3f;
        sweep_pace_feedback(us, n, blocks, 4.0f, 1);         // ... and its feedback, as the pipeline's fetcher gives it
        return 0;
    };

    uint64_t mag_mismatches = 0, sum_mismatches = 0, cand_mismatches = 0, ncand = 0;
    std::vector<float> cold;
    std::vector<uint16_t> hm(stride), hc((size_t) nsteps * kSwStep);
    std::vector<uint32_t> hn(nsteps + 1);
    std::vector<unsigned long long> hs((size_t) (buffers + 3) * 2);
    for (int rd = 0; rd < (queued ? 1 : rounds); ++rd) {
        for (int r = 0; r < replicas; ++r) {
            float us;
            if (run(r, us)) return 2;
            if (rd > 0) cold.push_back(us);                  // (the first round also pays the first touch of the code and the TLBs)
            if (rd == 0) {
                CK(hipMemcpy(hm.data(), d_mag + (size_t) r * stride, stride * 2, hipMemcpyDeviceToHost));
                for (uint64_t i = 0; i < n + kTrailing; ++i) mag_mismatches += hm[i] != want_mag[i];
                CK(hipMemcpy(hs.data(), d_sums, hs.size() * 8, hipMemcpyDeviceToHost));
                for (int b = 0; b < buffers; ++b) sum_mismatches += (hs[b] != want_level[b]) + (hs[buffers + 3 + b] != want_power[b]);
                CK(hipMemcpy(hc.data(), d_cand, hc.size() * 2, hipMemcpyDeviceToHost));
                CK(hipMemcpy(hn.data(), d_count, (size_t) (nsteps + 1) * 4, hipMemcpyDeviceToHost));
                std::vector<uint32_t> got;
                for (uint32_t s = 0; s < nsteps; ++s) {
                    if (hn[s] > (uint32_t) kSwStep) { ++cand_mismatches; continue; }
                    for (uint32_t i = 0; i < hn[s]; ++i) {
                        const uint32_t code = hc[(size_t) s * kSwStep + i];
                        const uint64_t pos = ((uint64_t) s * kSwStep & ~(uint64_t) (kUnit - 1)) + (code >> 3);
                        got.push_back((uint32_t) (pos << 3) | (code & 7u));
                    }
                }
                if (got.size() != want.size()) cand_mismatches += 1 + (got.size() > want.size() ? got.size() - want.size() : want.size() - got.size());
                else for (size_t i = 0; i < got.size(); ++i) cand_mismatches += got[i] != want[i];
                ncand = got.size();
            }
        }
    }
    if (queued) {
        // the timed rounds back to back on the stream, an event pair around every launch; the pace estimate moves once per round
        const int nl = (rounds - 1) * replicas;
        std::vector<hipEvent_t> ev(2 * (size_t) nl);
        for (auto &e : ev) CK(hipEventCreate(&e));
        for (int rd = 1; rd < rounds; ++rd) {
            for (int r = 0; r < replicas; ++r) {
                const int k = (rd - 1) * replicas + r;
                p.iq = d_iq + (size_t) r * n * 2 + iq_shift;
                p.mag = p.mag_w = d_mag + (size_t) r * stride;
                CK(hipMemsetAsync(d_dealer, 0, (size_t) 2 * kDealerCounters * kDealerStride * 4, nullptr));
                CK(hipMemsetAsync(d_sums, 0, (size_t) (buffers + 3) * 2 * 8, nullptr));
                CK(hipEventRecord(ev[2 * k], nullptr));
                blocks = launch_sweep(p, nullptr);
                CK(hipEventRecord(ev[2 * k + 1], nullptr));
            }
            CK(hipEventSynchronize(ev[2 * ((rd - 1) * replicas + replicas - 1) + 1]));
            for (int r = 0; r < replicas; ++r) {
                const int k = (rd - 1) * replicas + r;
                float ms = 0;
                CK(hipEventElapsedTime(&ms, ev[2 * k], ev[2 * k + 1]));
                cold.push_back(ms * 1e3f);
                sweep_pace_feedback(ms * 1e3f, n, blocks, 4.0f, 1);
            }
        }
        CK(hipGetLastError());
    }
    std::sort(cold.begin(), cold.end());
    double avg = 0;
    for (float x : cold) avg += x;
    avg /= cold.size();
    const double bytes = (double) n * 4.0;
    printf("{\"kernel\": \"k_sweep_uc8\", \"samples_per_launch\": %llu, \"algorithmic_bytes_per_launch\": %.0f, \"replicas\": %d, \"cold_bytes_walked\": %.0f, "
           "\"queued\": %d, \"blocks_of_256_threads\": %u, \"dense\": %d, \"rate\": %.0f, \"candidates\": %llu, \"candidates_cpu\": %zu, "
           "\"magnitude_mismatches_vs_cpu_table\": %llu, \"buffer_sum_mismatches\": %llu, \"candidate_mismatches_vs_cpu_scan\": %llu, "
           "\"us\": {\"min\": %.2f, \"median\": %.2f, \"mean\": %.2f, \"max\": %.2f, \"launches\": %zu, \"between\": \"HIP events around each launch (3.7 us of bracket included)\"}, "
           "\"GBs\": %.1f, \"frac_of_8TBs\": %.4f}\n",
           (unsigned long long) n, bytes, replicas, (double) replicas * ((double) n * 2 + (double) stride * 2), queued, blocks, dense, rate,
           (unsigned long long) ncand, want.size(), (unsigned long long) mag_mismatches, (unsigned long long) sum_mismatches, (unsigned long long) cand_mismatches,
           cold.front(), cold[cold.size() / 2], avg, cold.back(), cold.size(), bytes / (avg * 1e-6) / 1e9, bytes / (avg * 1e-6) / 1e9 / 8000.0);
    return (mag_mismatches || sum_mismatches || cand_mismatches) ? 1 : 0;
}
