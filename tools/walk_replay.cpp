// walk_replay.cpp — replays the ordered walk (readsb_amd/csrc/resolve.cpp) on a chunk of live records
// dumped from a GPU run (MGPU_DUMP_DIR), to profile the host side without a GPU.
//   g++ -O2 -std=c++17 -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ -o /tmp/walk_replay tools/walk_replay.cpp readsb_amd/csrc/resolve.cpp
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <thread>
#include <vector>
#include "../readsb_amd/csrc/resolve.h"
using namespace mgpu;
template <class T> static std::vector<T> rd(const std::string &p) {
    FILE *f = fopen(p.c_str(), "rb"); if (!f) { perror(p.c_str()); exit(1); }
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    std::vector<T> v(n / sizeof(T)); if (fread(v.data(), sizeof(T), v.size(), f) != v.size()) exit(1); fclose(f); return v;
}
int main(int argc, char **argv) {
    std::string d = argc > 1 ? argv[1] : "gpurun_out";
    auto recs = rd<PhaseRec>(d + "/walk_recs.bin");
    { PhaseRec s{}; s.pos = 0xFFFFFFFFu; recs.push_back(s); }   // sentinel the walk expects at recs[nrecs]
    const size_t nrecs = recs.size() - 1;
    auto sig = rd<unsigned long long>(d + "/walk_sig.bin");
    auto bufs = rd<BufferClock>(d + "/walk_bufs.bin");
    std::vector<uint32_t> pos(nrecs), lim(nrecs);
    std::vector<uint16_t> skip(nrecs);
    double best_d = 1e9, best_b = 1e9; size_t nm = 0;
    for (int it = 0; it < 20; ++it) {
        Resolver r; r.reset(1000000);
        std::vector<Accepted> acc; acc.reserve(nrecs);
        std::vector<mgpu_msg> out(nrecs + 16);
        ResolveCounts rc;
        auto t0 = std::chrono::steady_clock::now();
        int64_t n = r.decide(recs.data(), nrecs, bufs, acc, pos.data(), skip.data(), lim.data(), nrecs, rc);
        auto t1 = std::chrono::steady_clock::now();
        Resolver::build_messages(recs.data(), sig.data(), nullptr, bufs, acc.data(), (uint64_t) n, out.data());
        auto t2 = std::chrono::steady_clock::now();
        best_d = std::min(best_d, std::chrono::duration<double, std::milli>(t1 - t0).count());
        best_b = std::min(best_b, std::chrono::duration<double, std::milli>(t2 - t1).count());
        nm = (size_t) n;
    }
    printf("%zu records, %zu buffers -> %zu messages: decide %.3f ms (%.1f ns/record), build %.3f ms (%.1f ns/message)\n", nrecs, bufs.size(), nm,
           best_d, best_d * 1e6 / nrecs, best_b, best_b * 1e6 / nm);
    // ---- the speculative parallel walk against the serial one (same chunk walked twice: cold filter, then warm) ----
    const int K = argc > 2 ? atoi(argv[2]) : 4;
    {
        Resolver rs, rp; rs.reset(1000000); rp.reset(1000000);
        std::vector<SegmentWalk> seg(K);
        for (int round = 0; round < 4; ++round) {
            std::vector<Accepted> acc_s; ResolveCounts rc_s;
            int64_t ns = rs.decide(recs.data(), nrecs, bufs, acc_s, pos.data(), skip.data(), lim.data(), nrecs, rc_s);
            const uint32_t nb = (uint32_t) bufs.size();
            for (int k = 0; k < K; ++k) {
                seg[k].b_lo = (uint32_t) ((uint64_t) nb * k / K); seg[k].b_hi = (uint32_t) ((uint64_t) nb * (k + 1) / K);
                seg[k].rec_lo = segment_first_record(recs.data(), nrecs, bufs[seg[k].b_lo].first);
            }
            for (int k = 0; k < K; ++k) seg[k].rec_hi = k + 1 < K ? seg[k + 1].rec_lo : nrecs;
            auto t0 = std::chrono::steady_clock::now();
            uint64_t batches = 0;
            rp.parallel_walk(recs.data(), nrecs, bufs, seg, [&](int ntasks, const std::function<void(int)> &task) {
                std::vector<std::thread> th;
                for (int i = 0; i < ntasks; ++i) th.emplace_back([&task, i] { task(i); });
                for (auto &x : th) x.join();
            }, &batches);
            auto t1 = std::chrono::steady_clock::now();
            size_t np = 0; int nspec = 0; bool same = true;
            for (int k = 0; k < K; ++k) {
                nspec += seg[k].speculated;
                for (uint64_t i = 0; i < seg[k].nacc; ++i, ++np)
                    if (np >= (size_t) ns || acc_s[np].rec != seg[k].acc[i].rec || acc_s[np].buffer != seg[k].acc[i].buffer || acc_s[np].score != seg[k].acc[i].score) same = false;
            }
            auto t2 = std::chrono::steady_clock::now();
            (void) batches;
            std::vector<uint32_t> us, up; rs.union_snapshot(us); rp.union_snapshot(up);
            printf("round %d: serial %lld msgs, parallel(%d) %zu msgs, identical %d, filter unions equal %d, flips %llu/%llu, speculated %d/%d, walk %.3f ms + commit %.3f ms\n",
                   round, (long long) ns, K, np, (int) (same && np == (size_t) ns), (int) (us == up), (unsigned long long) rs.nflips(), (unsigned long long) rp.nflips(), nspec, K,
                   std::chrono::duration<double, std::milli>(t1 - t0).count(), std::chrono::duration<double, std::milli>(t2 - t1).count());
        }
    }
    // ---- the device walk's algorithm restated on the host (Resolver::device_walk_model: per-buffer walks, fixed point over the
    //      table of first adds, premise check) against the serial walk: same chunk four times, cold filter first ----
    {
        Resolver rs, rm; rs.reset(1000000); rm.reset(1000000);
        for (int round = 0; round < 4; ++round) {
            std::vector<Accepted> acc_s, acc_m; ResolveCounts rc_s, rc_m;
            const int64_t ns = rs.decide(recs.data(), nrecs, bufs, acc_s, pos.data(), skip.data(), lim.data(), nrecs, rc_s);
            uint32_t walks = 0;
            auto t0 = std::chrono::steady_clock::now();
            int64_t nm2 = rm.device_walk_model(recs.data(), nrecs, bufs, acc_m, rc_m, 3, &walks);
            auto t1 = std::chrono::steady_clock::now();
            const bool by_model = nm2 >= 0;
            if (!by_model) nm2 = rm.decide(recs.data(), nrecs, bufs, acc_m, pos.data(), skip.data(), lim.data(), nrecs, rc_m);
            bool same = nm2 == ns && rm.same_state(rs);
            for (int64_t i = 0; same && i < ns; ++i)
                same = acc_s[(size_t) i].rec == acc_m[(size_t) i].rec && acc_s[(size_t) i].buffer == acc_m[(size_t) i].buffer && acc_s[(size_t) i].score == acc_m[(size_t) i].score;
            printf("model round %d: serial %lld msgs, model %lld msgs, identical %d, decided by the model %d, walks %u, %.3f ms\n", round, (long long) ns,
                   (long long) nm2, (int) same, (int) by_model, walks, std::chrono::duration<double, std::milli>(t1 - t0).count());
        }
    }
    return 0;
}
