// walk_replay.cpp — replays the ordered walk (readsb_amd/csrc/resolve.cpp) on a chunk of live records
// dumped from a GPU run (MGPU_DUMP_DIR), to profile the host side without a GPU.
//   g++ -O2 -std=c++17 -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ -o /tmp/walk_replay tools/walk_replay.cpp readsb_amd/csrc/resolve.cpp
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
    auto sig = rd<unsigned long long>(d + "/walk_sig.bin");
    auto bufs = rd<BufferClock>(d + "/walk_bufs.bin");
    std::vector<uint32_t> pos(recs.size()), lim(recs.size());
    std::vector<uint16_t> skip(recs.size());
    double best = 1e9; size_t nm = 0;
    for (int it = 0; it < 20; ++it) {
        Resolver r; r.reset(1000000);
        std::vector<mgpu_msg> out; out.reserve(recs.size() / 4);
        ResolveCounts rc;
        auto t0 = std::chrono::steady_clock::now();
        int64_t n = r.walk(recs.data(), sig.data(), recs.size(), bufs, out, pos.data(), skip.data(), lim.data(), recs.size(), rc);
        double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (ms < best) best = ms;
        nm = (size_t) n;
    }
    printf("%zu records, %zu buffers -> %zu messages, best %.3f ms (%.1f ns/record)\n", recs.size(), bufs.size(), nm, best, best * 1e6 / recs.size());
    return 0;
}
