// selftest.cpp — host-logic self-checks that need no GPU (include/modes_gpu.h, "diagnostics"): the parallel walk, the device walk's
// algorithm restated on the host, and the sharded walk's whole protocol, each on seeded synthetic record streams against the serial
// walk.  No HIP in here: tools/sanitize_host.sh builds this file with resolve.cpp and seqsum.cpp under TSan and ASan + UBSan.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <functional>
#include <thread>
#include <vector>

#include "../../include/modes_gpu.h"
#include "resolve.h"

using namespace mgpu;

// Seeded synthetic record streams for the host-logic self-checks: aircraft that transmit during [on, off) of every `period`
// seconds — quiet spells longer than two filter generations make addresses expire, short ones do not — and the record kinds the
// kernels emit (clean adders, repaired frames, address-parity replies that only count when the address is known).
struct SelftestStream {
    struct Plane { uint32_t addr; double on, off, period, until; };
    uint64_t x;
    std::vector<Plane> planes;
    uint64_t rnd() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; }
    // front_extra more aircraft transmit during the first front_s seconds only: the filter's table grows for them and shrinks by one
    // size per expiry afterwards (icao_filter.c:96-110) — state that a rank starting later cannot rebuild from a warm-up
    SelftestStream(uint64_t seed, uint32_t naircraft, uint32_t front_extra = 0, double front_s = 0) : x(seed ? seed : 88172645463325252ull), planes(naircraft + front_extra) {
        for (uint32_t i = 0; i < naircraft + front_extra; ++i) {
            planes[i].until = i < naircraft ? 1e300 : front_s;
            planes[i].addr = 0x400000u + (uint32_t) (rnd() % 4096) * 7u + i;
            planes[i].period = 40.0 + (double) (rnd() % 400);
            planes[i].on = (double) (rnd() % 1000) / 1000.0 * planes[i].period;
            planes[i].off = planes[i].on + 5.0 + (double) (rnd() % 1000) / 1000.0 * planes[i].period;
            if (i >= naircraft) { planes[i].on = 0; planes[i].off = planes[i].period; }
        }
    }
    // one chunk of `nbuf` 131072-sample buffers starting at stream position `stream_pos`: its buffer grid and its records
    // (position order, no sentinel); returns the chunk's length in samples
    uint64_t chunk(uint64_t stream_pos, uint32_t nbuf, std::vector<BufferClock> &bufs, std::vector<PhaseRec> &recs) {
        const uint32_t B = 131072;
        const uint32_t naircraft = (uint32_t) planes.size();
        bufs.clear(); recs.clear();
        for (uint32_t b = 0; b < nbuf; ++b) {
            BufferClock bc;
            bc.first = b * B; bc.length = B;
            bc.sampleTimestamp = (int64_t) (stream_pos + (uint64_t) b * B) * 5;
            bc.sysTimestamp = bc.sampleTimestamp / 12000 + 1000000;
            bufs.push_back(bc);
        }
        const uint64_t npos = (uint64_t) nbuf * B;
        for (uint64_t pos = rnd() % 600; pos < npos; pos += 40 + rnd() % 900) {
            const double t = (double) (stream_pos + pos) / 2.4e6;
            const Plane &pl = planes[rnd() % naircraft];
            const double ph = std::fmod(t, pl.period);
            const bool active = ph >= pl.on && ph < pl.off && t < pl.until;
            const int nrec = 1 + (int) (rnd() % 3);
            int phase = 4 + (int) (rnd() % 3);
            for (int k = 0; k < nrec && phase <= 8; ++k, phase += 1 + (int) (rnd() % 2)) {
                PhaseRec r{};
                r.pos = (uint32_t) pos; r.phase = (uint8_t) phase; r.fixbit0 = r.fixbit1 = 0xff;
                const uint32_t kind = (uint32_t) (rnd() % 100);
                const uint32_t other = planes[rnd() % naircraft].addr;
                if (active && kind < 45) { r.flags = REC_ACCEPT_IF_UNKNOWN | REC_ADDER | REC_LONG; r.score_known = 1800; r.score_unknown = 1400; r.addr = pl.addr; r.msg[0] = 0x8d; }
                else if (active && kind < 55) { r.flags = REC_ACCEPT_IF_UNKNOWN | REC_ADDER; r.score_known = 1600; r.score_unknown = 750; r.addr = pl.addr; r.msg[0] = 0x5d; }
                else if (kind < 65) { r.flags = REC_LONG | (1u << REC_CORR_SHIFT); r.score_known = 900; r.score_unknown = 700; r.addr = other; r.msg[0] = 0x8d; r.fixbit0 = 40; }
                else if (kind < 80) { r.flags = REC_COND | REC_LONG; r.score_known = 1000; r.score_unknown = -1; r.addr = other; r.msg[0] = 0xa0; }
                else if (kind < 90) { r.flags = REC_COND; r.score_known = 1000; r.score_unknown = -1; r.addr = pl.addr; r.msg[0] = 0x20; }
                else { r.flags = REC_COND | (1u << REC_CORR_SHIFT); r.score_known = 800; r.score_unknown = -1; r.addr = other; r.msg[0] = 0x5d; r.fixbit0 = 12; }
                recs.push_back(r);
            }
        }
        return npos;
    }
};

extern "C" {

// The device walk's algorithm (Resolver::device_walk_model = kernels/walk.inc restated on the host) against the serial walk on the
// same streams: every decision, every counter, the filter afterwards.  Chunks whose premises fail (the table grows inside the
// chunk) or that do not settle are walked serially, as the library does.  0 = identical, k > 0 = first differing chunk + 1.
// stats[0] chunks decided by the model, [1] chunks walked serially after all, [2] walks in all, [3] most walks one chunk took.
int mgpu_selftest_device_walk(uint64_t seed, uint32_t nchunks, uint32_t buffers_per_chunk, uint32_t naircraft, uint32_t max_walks, uint64_t stats[4]) {
    if (nchunks == 0 || buffers_per_chunk == 0 || naircraft == 0 || max_walks == 0) return -1;
    SelftestStream gen(seed, naircraft);
    Resolver serial, model;
    serial.reset(1000000);
    model.reset(1000000);
    uint64_t st[4] = {0, 0, 0, 0};
    uint64_t stream_pos = 0;
    for (uint32_t ch = 0; ch < nchunks; ++ch) {
        std::vector<BufferClock> bufs;
        std::vector<PhaseRec> recs;
        const uint64_t npos = gen.chunk(stream_pos, buffers_per_chunk, bufs, recs);
        const uint64_t n = recs.size();
        { PhaseRec s{}; s.pos = 0xFFFFFFFFu; recs.push_back(s); }
        std::vector<uint32_t> pos(n + 1), lim(n + 1);
        std::vector<uint16_t> skip(n + 1);
        std::vector<Accepted> acc_s, acc_m;
        ResolveCounts rc_s, rc_m;
        const int64_t ns = serial.decide(recs.data(), n, bufs, acc_s, pos.data(), skip.data(), lim.data(), n + 1, rc_s);
        uint32_t walks = 0;
        int64_t nm = model.device_walk_model(recs.data(), n, bufs, acc_m, rc_m, max_walks, &walks);
        st[2] += walks;
        if (walks > st[3]) st[3] = walks;
        if (nm < 0) { ++st[1]; nm = model.decide(recs.data(), n, bufs, acc_m, pos.data(), skip.data(), lim.data(), n + 1, rc_m); }
        else ++st[0];
        bool same = nm == ns && std::memcmp(&rc_s, &rc_m, sizeof(rc_s)) == 0 && model.same_state(serial);
        for (int64_t i = 0; same && i < ns; ++i)
            same = acc_s[(size_t) i].rec == acc_m[(size_t) i].rec && acc_s[(size_t) i].buffer == acc_m[(size_t) i].buffer && acc_s[(size_t) i].score == acc_m[(size_t) i].score;
        if (!same) { if (stats) std::memcpy(stats, st, sizeof(st)); return (int) ch + 1; }
        stream_pos += npos;
    }
    if (stats) std::memcpy(stats, st, sizeof(st));
    return 0;
}

// Host-logic self-check (no GPU): a seeded synthetic record stream — aircraft that appear, go quiet and
// return, so that addresses enter the ICAO filter, expire on the 60 s clock and come back — is walked
// chunk by chunk once with Resolver::decide and once with Resolver::parallel_walk over `nsegments`
// buffer ranges per chunk (real threads).  Returns 0 when every decision, every counter and the final
// filter agree; k > 0 = first differing chunk + 1.  *speculated_permille = share of chunks whose ranges
// all committed in the first batch (a test that never restarted a batch, or always did, would prove little).
int mgpu_selftest_walk(uint64_t seed, uint32_t nchunks, uint32_t buffers_per_chunk, uint32_t nsegments, uint32_t naircraft,
                       uint32_t *speculated_permille) {
    if (nchunks == 0 || buffers_per_chunk == 0 || nsegments == 0 || naircraft == 0) return -1;
    SelftestStream gen(seed, naircraft);
    Resolver serial, parallel;
    serial.reset(1000000);
    parallel.reset(1000000);
    uint64_t held = 0, ranges = 0;   // chunks that took a single batch / chunks
    const uint32_t K = nsegments < buffers_per_chunk ? nsegments : buffers_per_chunk;
    std::vector<SegmentWalk> segs(K);
    uint64_t stream_pos = 0;
    for (uint32_t ch = 0; ch < nchunks; ++ch) {
        std::vector<BufferClock> bufs;
        std::vector<PhaseRec> recs;
        const uint64_t npos = gen.chunk(stream_pos, buffers_per_chunk, bufs, recs);
        const uint64_t n = recs.size();
        { PhaseRec s{}; s.pos = 0xFFFFFFFFu; recs.push_back(s); }
        std::vector<uint32_t> pos(n + 1), lim(n + 1);
        std::vector<uint16_t> skip(n + 1);
        std::vector<Accepted> acc_s;
        ResolveCounts rc_s, rc_p;
        const int64_t ns = serial.decide(recs.data(), n, bufs, acc_s, pos.data(), skip.data(), lim.data(), n + 1, rc_s);
        for (uint32_t k = 0; k < K; ++k) {
            segs[k].b_lo = (uint32_t) ((uint64_t) buffers_per_chunk * k / K);
            segs[k].b_hi = (uint32_t) ((uint64_t) buffers_per_chunk * (k + 1) / K);
            segs[k].rec_lo = k == 0 ? 0 : segment_first_record(recs.data(), n, bufs[segs[k].b_lo].first);
        }
        for (uint32_t k = 0; k < K; ++k) segs[k].rec_hi = k + 1 < K ? segs[k + 1].rec_lo : n;
        uint64_t batches = 0;
        parallel.parallel_walk(recs.data(), n, bufs, segs, [&](int ntasks, const std::function<void(int)> &task) {
            std::vector<std::thread> th;
            for (int i = 0; i < ntasks; ++i) th.emplace_back([&task, i] { task(i); });
            for (auto &t : th) t.join();
        }, &batches);
        held += batches == 1 ? 1 : 0;
        ++ranges;
        uint64_t np = 0;
        bool same = true;
        for (uint32_t k = 0; k < K; ++k) {
            rc_p.add(segs[k].counts);
            for (uint64_t i = 0; i < segs[k].nacc; ++i, ++np) {
                if (np >= (uint64_t) ns) { same = false; break; }
                const Accepted &p = segs[k].acc[i], &s = acc_s[np];
                if (p.rec != s.rec || p.buffer != s.buffer || p.score != s.score || segs[k].pos[i] != pos[np] ||
                    segs[k].skip[i] != skip[np] || segs[k].limit[i] != lim[np]) same = false;
            }
        }
        std::vector<uint32_t> us, up;
        serial.union_snapshot(us);
        parallel.union_snapshot(up);
        if (!same || np != (uint64_t) ns || std::memcmp(&rc_s, &rc_p, sizeof(rc_s)) != 0 || us != up ||
            serial.nflips() != parallel.nflips() || serial.filter().occupied() != parallel.filter().occupied() ||
            serial.filter().table_bits() != parallel.filter().table_bits())
            return (int) ch + 1;
        stream_pos += npos;
    }
    if (speculated_permille) *speculated_permille = ranges ? (uint32_t) (held * 1000 / ranges) : 0;
    return 0;
}

// The sharded walk's protocol (mgpu_shard_walk, readsb_amd/shard.py) on a synthetic record stream, without a GPU: the capture's
// chunks are dealt to `nranks` ranks (whole chunks, contiguous), every rank walks warm-up + range from an empty filter with the
// schedule derived from ESTIMATED end clocks imposed, and the fixed point over (schedule, seam states) is iterated exactly as the
// ranks would with all-gathers in between.  Against the serial walk of the whole stream: every decision of every chunk (from the
// rank that owns it), every buffer's end clock, the counts, the number of expiries, the last rank's final state.
// 0 = identical; k > 0 = first differing chunk + 1; -2 = the iteration did not settle.  stats: [0] protocol rounds, [1] walks of a
// range in all, [2] seams that failed in some round, [3] rounds in which the schedule changed, [4] expiries, [5] ranges that started
// from an imported state in the end.  flags bit 0: the first schedule from the buffers' START clocks instead of the estimate;
// bit 1: a deliberately wrong first schedule (the rounds over the schedule have work to do).
int mgpu_selftest_shard_walk(uint64_t seed, uint32_t nchunks, uint32_t buffers_per_chunk, uint32_t naircraft, uint32_t front_extra,
                             uint32_t nranks, uint32_t nsegments, uint32_t flags, uint64_t stats[6]) {
    if (nchunks == 0 || buffers_per_chunk == 0 || naircraft == 0 || nranks == 0 || nranks > nchunks) return -1;
    const uint32_t B = 131072;
    const int64_t startup = 1000000;
    struct Chunk { std::vector<BufferClock> bufs; std::vector<PhaseRec> recs; uint64_t pos, n; uint64_t nrecs; };
    std::vector<Chunk> chunks(nchunks);
    {
        SelftestStream gen(seed, naircraft, front_extra, 150.0);
        uint64_t pos = 0;
        for (auto &ch : chunks) {
            ch.pos = pos;
            ch.n = gen.chunk(pos, buffers_per_chunk, ch.bufs, ch.recs);
            ch.nrecs = ch.recs.size();
            PhaseRec sentinel{}; sentinel.pos = 0xFFFFFFFFu; ch.recs.push_back(sentinel);
            pos += ch.n;
        }
    }
    struct Decisions { std::vector<Accepted> acc; ResolveCounts rc; };
    auto walk_chunk = [&](Resolver &res, const Chunk &ch, Decisions &d, bool parallel) {
        const uint64_t n = ch.nrecs;
        std::vector<uint32_t> pos(n + 1), lim(n + 1);
        std::vector<uint16_t> skip(n + 1);
        d.rc = ResolveCounts();
        const uint32_t K = nsegments < buffers_per_chunk ? nsegments : buffers_per_chunk;
        if (!parallel || K < 2) {
            const int64_t na = res.decide(ch.recs.data(), n, ch.bufs, d.acc, pos.data(), skip.data(), lim.data(), n + 1, d.rc);
            d.acc.resize((size_t) (na < 0 ? 0 : na));
            return;
        }
        std::vector<SegmentWalk> segs(K);
        for (uint32_t k = 0; k < K; ++k) {
            segs[k].b_lo = (uint32_t) ((uint64_t) buffers_per_chunk * k / K);
            segs[k].b_hi = (uint32_t) ((uint64_t) buffers_per_chunk * (k + 1) / K);
            segs[k].rec_lo = k == 0 ? 0 : segment_first_record(ch.recs.data(), n, ch.bufs[segs[k].b_lo].first);
        }
        for (uint32_t k = 0; k < K; ++k) segs[k].rec_hi = k + 1 < K ? segs[k + 1].rec_lo : n;
        res.parallel_walk(ch.recs.data(), n, ch.bufs, segs, [&](int ntasks, const std::function<void(int)> &task) {
            std::vector<std::thread> th;
            for (int i = 0; i < ntasks; ++i) th.emplace_back([&task, i] { task(i); });
            for (auto &t : th) t.join();
        });
        d.acc.clear();
        for (uint32_t k = 0; k < K; ++k) {
            d.rc.add(segs[k].counts);
            d.acc.insert(d.acc.end(), segs[k].acc.begin(), segs[k].acc.begin() + (std::ptrdiff_t) segs[k].nacc);
        }
    };
    // ---- the serial walk of the whole stream ----
    Resolver truth;
    truth.reset(startup, 0);
    std::vector<int64_t> true_clocks;
    truth.log_end_clocks(&true_clocks);
    std::vector<Decisions> want(nchunks);
    for (uint32_t i = 0; i < nchunks; ++i) walk_chunk(truth, chunks[i], want[i], false);
    truth.log_end_clocks(nullptr);
    std::vector<uint8_t> true_end;
    truth.export_state(true_end);
    // ---- the ranks ----
    const double chunk_s = (double) buffers_per_chunk * B / 2.4e6;
    const uint32_t warm = (uint32_t) (120.3 / chunk_s) + 2;                    // two generations (+ the expiry's slack) in whole chunks, one to spare
    struct Rank { uint32_t c0, c1, w0; Resolver res; ShardWalkOut out; std::vector<Decisions> got; std::vector<uint8_t> import; bool walked = false; };
    std::vector<Rank> ranks(nranks);
    for (uint32_t r = 0; r < nranks; ++r) {
        ranks[r].c0 = (uint32_t) ((uint64_t) nchunks * r / nranks);
        ranks[r].c1 = (uint32_t) ((uint64_t) nchunks * (r + 1) / nranks);
        ranks[r].w0 = ranks[r].c0 > warm ? ranks[r].c0 - warm : 0;
        ranks[r].got.resize(ranks[r].c1 - ranks[r].c0);
    }
    std::vector<int64_t> clocks;
    for (uint32_t i = 0; i < nchunks; ++i) {
        if (flags & 1u) for (const BufferClock &b : chunks[i].bufs) clocks.push_back(b.sysTimestamp);   // a crude first guess: the iteration over the schedule has work to do
        else estimate_end_clocks(chunks[i].recs.data(), chunks[i].nrecs, chunks[i].bufs, clocks);
    }
    std::vector<uint64_t> fl;
    flip_schedule(clocks.data(), clocks.size(), startup, 0, fl);
    std::vector<int64_t> sched(fl.size());
    for (size_t i = 0; i < fl.size(); ++i) sched[i] = (int64_t) (fl[i] * B) * 5;
    if (flags & 2u) for (size_t i = 1; i < sched.size(); i += 2) sched[i] += (int64_t) B * 5 * (int64_t) (1 + i % 3);   // a WRONG first schedule: some expiries 1-3 buffers late
    uint64_t st[6] = {0, 0, 0, 0, 0, 0};
    bool done = false;
    std::vector<int64_t> used_sched;
    for (uint32_t round = 0; round < nranks + 70 && !done; ++round) {
        ++st[0];
        const bool sched_changed = used_sched != sched;
        used_sched = sched;
        for (uint32_t r = 0; r < nranks; ++r) {
            Rank &R = ranks[r];
            if (R.walked && !sched_changed && (R.import.empty() || R.import == R.out.state_first)) continue;   // nothing it depends on has changed
            ShardWalkPlan plan;
            plan.own_first = chunks[R.c0].pos; plan.buf_samples = B; plan.startup_ms = startup; plan.clock_mode = 0;
            plan.sched = used_sched.data(); plan.nsched = used_sched.size();
            if (!R.import.empty()) { plan.start_state = R.import.data(); plan.start_state_bytes = R.import.size(); }
            const char *err = "";
            const int rc = shard_walk_core(R.res, plan, R.c1 - R.w0,
                [&](size_t i, uint64_t &pos, uint64_t &n) { pos = chunks[R.w0 + i].pos; n = chunks[R.w0 + i].n; },
                [&](size_t i, bool own) { Decisions scratch; walk_chunk(R.res, chunks[R.w0 + i], own ? R.got[R.w0 + i - R.c0] : scratch, true); return 0; },
                R.out, &err);
            if (rc != 0) { fprintf(stderr, "mgpu_selftest_shard_walk: rank %u: %s\n", r, err); return -1; }
            R.walked = true;
            ++st[1];
        }
        // ---- what the all-gather would hand every rank: all clocks, all states ----
        clocks.clear();
        for (auto &R : ranks) clocks.insert(clocks.end(), R.out.clocks.begin(), R.out.clocks.end());
        flip_schedule(clocks.data(), clocks.size(), startup, 0, fl);
        std::vector<int64_t> next(fl.size());
        for (size_t i = 0; i < fl.size(); ++i) next[i] = (int64_t) (fl[i] * B) * 5;
        bool seams = true;
        for (uint32_t r = 1; r < nranks; ++r)
            if (ranks[r].out.state_first != ranks[r - 1].out.state_end) { seams = false; ++st[2]; ranks[r].import = ranks[r - 1].out.state_end; }
        if (next != sched) ++st[3];
        done = seams && next == sched;
        sched.swap(next);
    }
    if (stats) { st[4] = sched.size(); for (auto &R : ranks) st[5] += R.import.empty() ? 0 : 1; std::memcpy(stats, st, sizeof(st)); }
    if (!done) return -2;
    // ---- against the serial walk ----
    if (clocks != true_clocks) return (int) nchunks + 1;
    for (uint32_t r = 0; r < nranks; ++r)
        for (uint32_t i = ranks[r].c0; i < ranks[r].c1; ++i) {
            const Decisions &g = ranks[r].got[i - ranks[r].c0], &w = want[i];
            bool same = g.acc.size() == w.acc.size() && std::memcmp(&g.rc, &w.rc, sizeof(g.rc)) == 0;
            for (size_t k = 0; same && k < g.acc.size(); ++k) same = g.acc[k].rec == w.acc[k].rec && g.acc[k].buffer == w.acc[k].buffer && g.acc[k].score == w.acc[k].score;
            if (!same) return (int) i + 1;
        }
    if (ranks[nranks - 1].out.state_end != true_end) return (int) nchunks + 2;
    return 0;
}

}  // extern "C"
