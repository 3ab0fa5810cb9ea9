// walk_replay.cpp — replays the ordered walk (readsb_amd/csrc/resolve.cpp) on a chunk of live records
// dumped from a GPU run (MGPU_DUMP_DIR), to profile the host side without a GPU.
//   g++ -O2 -std=c++17 -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ -o /tmp/walk_replay tools/walk_replay.cpp readsb_amd/csrc/resolve.cpp
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
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
        Resolver::build_messages(recs.data(), sig.data(), bufs, acc.data(), (uint64_t) n, out.data());
        auto t2 = std::chrono::steady_clock::now();
        best_d = std::min(best_d, std::chrono::duration<double, std::milli>(t1 - t0).count());
        best_b = std::min(best_b, std::chrono::duration<double, std::milli>(t2 - t1).count());
        nm = (size_t) n;
    }
    printf("%zu records, %zu buffers -> %zu messages: decide %.3f ms (%.1f ns/record), build %.3f ms (%.1f ns/message)\n", nrecs, bufs.size(), nm,
           best_d, best_d * 1e6 / nrecs, best_b, best_b * 1e6 / nm);
    return 0;
}
