// tools/micro/sweep_cold.hip — k_sweep (the kernel the HBM roofline applies to) on its own, over a COLD magnitude array.
//
// In the pipeline k_sweep reads the 134 MB of magnitudes k_convert_uc8 wrote microseconds earlier, and the MI355X has 256 MiB of
// Infinity Cache in front of HBM (FETCH_SIZE counts its hits too): the in-pipeline figure does not show that the bytes came from
// HBM.  Here the chunk's magnitudes are replicated R times (R x 134 MB > 1 GiB) and the launches walk the replicas round-robin, so
// every launch reads memory that was last touched R - 1 launches (> 1 GB of traffic) ago.  The candidate lists of every launch are
// compared with a plain CPU scan of the same magnitudes (demod_2400.c:311-378 restated in check_cpu below).
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DMGPU_EXPERIMENTS=1 [-DMGPU_SW_STAGE=k] -o sweep_cold sweep_cold.hip -ldl
//   sweep_cold [buffers=512] [replicas=9] [rounds=4] [dense=0] [rate=2000] [blocks=0 (resident grid)] [ragged=0 (samples added to the chunk)] [pace=-1 (feedback as the library's; 0 off; else 10 ns ticks per step)]
// Output: one JSON line (per-launch time cold / warm, GB/s of algorithmic bytes, fraction of the 8 TB/s peak, wave lifetimes).
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

// demodulate2400's pre-check + threshold tests for every position of mag[0 .. n + 326) -> (position, phase mask)
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
    const int buffers = argc > 1 ? atoi(argv[1]) : 512;
    const int replicas = argc > 2 ? atoi(argv[2]) : 9;
    const int rounds = argc > 3 ? atoi(argv[3]) : 4;
    const int dense = argc > 4 ? atoi(argv[4]) : 0;
    const double rate = argc > 5 ? atof(argv[5]) : 2000.0;
    unsigned blocks = argc > 6 ? (unsigned) atoi(argv[6]) : 0u;
    const int ragged = argc > 7 ? atoi(argv[7]) : 0;
    const int pace_arg = argc > 8 ? atoi(argv[8]) : -1;      // -1: paced with the library's feedback rule, 0: no pacing, else fixed ticks
    float pace_ticks = pace_arg < 0 ? 260.0f : (float) pace_arg;
    const int thr = 58;
    const uint64_t n = (uint64_t) buffers * 131072 + (uint64_t) ragged;
    const uint64_t stride = ((n + kTrailing + 4096 + 4095) / 4096) * 4096;      // magnitudes per replica (16-byte aligned, with the tile slack)

    std::string here = argv[0];
    here = here.substr(0, here.find_last_of('/') == std::string::npos ? 0 : here.find_last_of('/'));
    const std::string so = (here.empty() ? std::string(".") : here) + "/../libsynth_iq.so";
    void *h = dlopen(so.c_str(), RTLD_NOW);
    if (!h) { fprintf(stderr, "%s: %s\n", so.c_str(), dlerror()); return 2; }
    synth_fn synth = (synth_fn) dlsym(h, "synth_iq_generate");
    std::vector<uint8_t> iq(n * 2);
    synth(424242, 0, rate, 200, dense, 3.0, 0, n, iq.data(), 32);

    uint8_t *d_iq;
    uint16_t *d_mag, *d_cand, *d_lut;
    uint32_t *d_count, *d_part, *d_dealer;
    unsigned long long *d_sums, *d_waves;
    const uint32_t nsteps = (uint32_t) ((n + kSwStep - 1) / kSwStep);
    CK(hipMalloc(&d_iq, n * 2));
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
    CK(hipMemcpy(d_iq, iq.data(), n * 2, hipMemcpyHostToDevice));
    CK(hipMemset(d_mag, 0, (size_t) replicas * stride * 2));
    CK(hipMemset(d_sums, 0, (size_t) (buffers + 3) * 2 * 8));
    ConvertParams cp{};
    cp.iq = d_iq; cp.mag = d_mag; cp.n = n; cp.buf_samples = 131072; cp.tail = nullptr; cp.uc8_folded = d_lut;
    cp.sum_level = d_sums; cp.sum_power = d_sums + buffers + 3;
    launch_convert(0, cp, nullptr);
    CK(hipDeviceSynchronize());
    for (int r = 1; r < replicas; ++r) CK(hipMemcpy(d_mag + (size_t) r * stride, d_mag, stride * 2, hipMemcpyDeviceToDevice));
    std::vector<uint16_t> mag(stride);
    CK(hipMemcpy(mag.data(), d_mag, stride * 2, hipMemcpyDeviceToHost));
    std::vector<uint32_t> want;
    check_cpu(mag.data(), n, thr, want);

    if (!blocks) blocks = resident_grid((const void *) k_sweep, 0, kSwMaxBlocks);
    const unsigned want_blocks = (nsteps + 3) / 4;
    if (blocks > want_blocks) blocks = want_blocks;
    if (blocks > 65536) blocks = 65536;
    SweepParams p{};
    p.n = n; p.thr = thr; p.cand = d_cand; p.cand_count = d_count; p.sweep_part = d_part; p.dbg_waves = d_waves; p.dealer = d_dealer;

    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto run = [&](int r, float &us) -> int {
        p.mag = d_mag + (size_t) r * stride;
        CK(hipMemsetAsync(d_dealer, 0, (size_t) 2 * kDealerCounters * kDealerStride * 4, nullptr));   // (the pipeline's k_publish hands the counters back zeroed)
        CK(hipEventRecord(e0, nullptr));
        p.pace_recip = pace_ticks > 0.0f ? (uint32_t) (4294967296.0 / pace_ticks) : 0u;
        hipLaunchKernelGGL(k_sweep, dim3(blocks), dim3(kBlock), 0, nullptr, p);
        CK(hipEventRecord(e1, nullptr));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        us = ms * 1e3f;
        if (pace_arg < 0 && (double) nsteps / (blocks * 4.0) >= 4.0 && us > 8.0f) {   // the library's feedback (sweep_pace_feedback)
            float t = (float) ((us - 4.5) * 100.0 / ((double) nsteps / (blocks * 4.0)));
            t = t < 120.0f ? 120.0f : t > 800.0f ? 800.0f : t;
            pace_ticks = 0.5f * pace_ticks + 0.5f * t;
        }
        return 0;
    };
    // ---- correctness of every replica's lists against the CPU scan (first round) ----
    uint64_t mismatches = 0, ncand = 0;
    std::vector<float> cold, warm;
    std::vector<uint16_t> hc((size_t) nsteps * kSwStep);
    std::vector<uint32_t> hn(nsteps + 1);
    for (int rd = 0; rd < rounds; ++rd) {
        for (int r = 0; r < replicas; ++r) {
            float us;
            if (run(r, us)) return 2;
            if (rd > 0) cold.push_back(us);                // (the first round also pays the first-touch of the code and the TLBs)
            if (rd == 0 && MGPU_SW_STAGE == 0 && MGPU_SW_EXP != 1 && MGPU_SW_EXP != 2 && (r == 0 || r == replicas - 1)) {
                CK(hipMemcpy(hc.data(), d_cand, hc.size() * 2, hipMemcpyDeviceToHost));
                CK(hipMemcpy(hn.data(), d_count, (size_t) (nsteps + 1) * 4, hipMemcpyDeviceToHost));
                std::vector<uint32_t> got;
                for (uint32_t s = 0; s < nsteps; ++s) {
                    if (hn[s] > (uint32_t) kSwStep) { ++mismatches; continue; }
                    for (uint32_t i = 0; i < hn[s]; ++i) {
                        const uint32_t code = hc[(size_t) s * kSwStep + i];
                        const uint64_t pos = ((uint64_t) s * kSwStep & ~(uint64_t) (kUnit - 1)) + (code >> 3);
                        got.push_back((uint32_t) (pos << 3) | (code & 7u));
                    }
                }
                if ((nsteps & 1u) && hn[nsteps] != 0) ++mismatches;
                if (got.size() != want.size()) mismatches += 1 + (got.size() > want.size() ? got.size() - want.size() : want.size() - got.size());
                else for (size_t i = 0; i < got.size(); ++i) mismatches += got[i] != want[i];
                ncand = got.size();
            }
        }
    }
    for (int i = 0; i < 2 * replicas; ++i) { float us; if (run(0, us)) return 2; if (i >= 2) warm.push_back(us); }
    // wave lifetimes of the last (warm) launch
    const unsigned nwaves = blocks * 4;
    std::vector<unsigned long long> wt((size_t) nwaves * 2);
    CK(hipMemcpy(wt.data(), d_waves, wt.size() * 8, hipMemcpyDeviceToHost));
    unsigned long long t0 = ~0ull, t1 = 0;
    double life = 0;
    std::vector<double> ends, starts;
    for (unsigned w = 0; w < nwaves; ++w) { t0 = std::min(t0, wt[2 * w]); t1 = std::max(t1, wt[2 * w + 1]); }
    for (unsigned w = 0; w < nwaves; ++w) {
        life += (double) (wt[2 * w + 1] - wt[2 * w]) / 100.0;
        starts.push_back((double) (wt[2 * w] - t0) / 100.0);
        ends.push_back((double) (wt[2 * w + 1] - t0) / 100.0);
    }
    std::sort(ends.begin(), ends.end());
    std::sort(starts.begin(), starts.end());
    auto stat = [](std::vector<float> v, double &mn, double &md, double &mx, double &avg) {
        std::sort(v.begin(), v.end());
        mn = v.front(); mx = v.back(); md = v[v.size() / 2]; avg = 0;
        for (float x : v) avg += x;
        avg /= v.size();
    };
    double cmin, cmed, cmax, cavg, wmin, wmed, wmax, wavg;
    stat(cold, cmin, cmed, cmax, cavg);
    stat(warm, wmin, wmed, wmax, wavg);
    const double bytes = (double) n * 2.0;
    printf("{\"kernel\": \"k_sweep\", \"variant\": \"exp %d nbuf %d deal %d touch %d\", \"stage\": %d, \"samples_per_launch\": %llu, \"algorithmic_bytes_per_launch\": %.0f, \"replicas\": %d, "
           "\"cold_array_bytes\": %.0f, \"pace_ticks\": %.0f, \"blocks\": %u, \"waves\": %u, \"dense\": %d, \"rate\": %.0f, \"candidates\": %llu, \"candidates_cpu\": %zu, "
           "\"mismatches_vs_cpu_scan\": %llu, "
           "\"cold_us\": {\"min\": %.2f, \"median\": %.2f, \"mean\": %.2f, \"max\": %.2f, \"launches\": %zu}, "
           "\"cold_GBs\": %.1f, \"cold_frac_of_8TBs\": %.4f, "
           "\"warm_us\": {\"min\": %.2f, \"median\": %.2f, \"mean\": %.2f, \"max\": %.2f, \"launches\": %zu}, \"warm_GBs\": %.1f, \"warm_frac_of_8TBs\": %.4f, "
           "\"waves_alive_frac\": %.3f, \"wave_life_mean_us\": %.2f, \"last_start_us\": %.2f, \"end_us\": {\"p10\": %.2f, \"p50\": %.2f, \"p90\": %.2f, \"max\": %.2f}}\n",
           (int) MGPU_SW_EXP, (int) MGPU_SW_NBUF, (int) MGPU_SW_DEAL, (int) MGPU_SW_TOUCH, (int) MGPU_SW_STAGE, (unsigned long long) n, bytes, replicas, (double) replicas * stride * 2, (double) pace_ticks, blocks, nwaves, dense, rate,
           (unsigned long long) ncand, want.size(), (unsigned long long) mismatches,
           cmin, cmed, cavg, cmax, cold.size(), bytes / (cavg * 1e-6) / 1e9, bytes / (cavg * 1e-6) / 1e9 / 8000.0,
           wmin, wmed, wavg, wmax, warm.size(), bytes / (wavg * 1e-6) / 1e9, bytes / (wavg * 1e-6) / 1e9 / 8000.0,
           life / nwaves / ((double) (t1 - t0) / 100.0), life / nwaves, starts.back(), ends[nwaves / 10], ends[nwaves / 2], ends[nwaves * 9 / 10], ends.back());
    return mismatches ? 1 : 0;
}
