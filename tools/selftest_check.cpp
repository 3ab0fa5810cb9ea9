// The library's GPU-free host logic under the sanitizers (tools/sanitize_host.sh): the speculative parallel walk, the sharded walk's
// whole protocol (warm-ups, imposed expiry schedule, rounds, state import) and the block-wise sequential double sum, each against its
// serial form.  Built from readsb_amd/csrc/{selftest,resolve,seqsum}.cpp with g++; nothing of HIP is linked.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../include/modes_gpu.h"

static int fails = 0;
static void expect(bool ok, const char *what) {
    printf("%-58s %s\n", what, ok ? "ok" : "FAILED");
    fails += !ok;
}

int main() {
    uint32_t spec = 0;
    expect(mgpu_selftest_walk(3, 12, 128, 8, 300, &spec) == 0, "parallel walk, 8 ranges per chunk");
    expect(mgpu_selftest_walk(5, 6, 512, 16, 2000, &spec) == 0, "parallel walk, 16 ranges per chunk");
    uint64_t ds[4] = {0};
    expect(mgpu_selftest_device_walk(7, 8, 128, 300, 6, ds) == 0, "device walk's fixed point (host restatement)");
    // the cases of tests/test_shard_walk.py that exercise each branch, at sizes a sanitized build walks in seconds
    struct { uint64_t seed; uint32_t chunks, bufs, ac, front, ranks, segs, flags; const char *what; } cases[] = {
        {1, 40, 128, 200, 0, 4, 4, 0, "sharded walk: plain"},
        {2, 40, 128, 200, 0, 8, 1, 0, "sharded walk: serial chunk walks"},
        {4, 30, 512, 1000, 0, 3, 8, 0, "sharded walk: a seam fails, the rank behind imports"},
        {1486393352ull, 57, 128, 3000, 3000, 8, 4, 0, "sharded walk: table-size hysteresis, several rounds"},
        {68494888361ull, 14, 256, 3000, 0, 3, 8, 3, "sharded walk: wrong first schedule corrected"},
    };
    for (auto &c : cases) {
        uint64_t st[6] = {0};
        int rc = mgpu_selftest_shard_walk(c.seed, c.chunks, c.bufs, c.ac, c.front, c.ranks, c.segs, c.flags, st);
        expect(rc == 0, c.what);
    }
    // block-wise sequential sum (threaded preparation) against the plain loop, across binade changes and with a wrong prediction
    std::vector<mgpu_msg> msgs(300000);
    uint64_t x = 88172645463325252ull;
    for (auto &m : msgs) {
        memset(&m, 0, sizeof m);
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        m.sig_sumsq = (x >> 40) % 4000000u + ((x & 1023) == 0 ? 4000000000u : 1u);
    }
    for (double start : {0.0, 1.0, 12345.678, 1e9}) {
        for (double approx : {start, start * 1.5 + 3}) {
            std::vector<mgpu_sum_block> blocks((msgs.size() + 1023) / 1024);
            mgpu_seqsum_blocks(approx, msgs.data(), msgs.size(), 1024, blocks.data());
            uint64_t fb = 0;
            double a = mgpu_seqsum_apply(start, msgs.data(), msgs.size(), 1024, blocks.data(), &fb);
            double b = mgpu_seqsum_signal_power(start, msgs.data(), msgs.size());
            char what[96];
            snprintf(what, sizeof what, "block sum, start %g predicted %g (%llu re-added)", start, approx, (unsigned long long) fb);
            expect(memcmp(&a, &b, 8) == 0, what);
        }
    }
    return fails ? 1 : 0;
}
