// resolve.cpp — ordered accept / skip-ahead / ICAO-filter walk (see resolve.h).
#include "resolve.h"

#include <algorithm>
#include <cstddef>
#include <cstring>
#include <limits>
#if defined(__SSE2__)
#include <emmintrin.h>
#endif

namespace mgpu {

static constexpr uint32_t kShowOnlyDefault = 0xff123456u;   // BADDR, readsb.h:296; modesInit adds it (readsb.c:310)
static constexpr int64_t kFilterTtlMs = 60000;              // MODES_ICAO_FILTER_TTL, readsb.h:315
static constexpr int64_t kBufferSpanMs = 64;                // > the clock span of one 131072-sample buffer (54.7 ms) + one frame

IcaoFilter::IcaoFilter() {
    for (int g = 0; g < 2; ++g) bits_[g].assign(1u << 18, 0);
    init();
}

void IcaoFilter::clear_gen(int g) {
    if (drops_)   // what leaves the union: members of this generation the other one does not hold
        for (uint32_t a : members_[g])
            if (!((bits_[g ^ 1][a >> 6] >> (a & 63)) & 1)) drops_->push_back(a);
    for (uint32_t a : members_[g]) bits_[g][a >> 6] = 0;
    members_[g].clear();
    big_[g].clear();
}

void IcaoFilter::init() {
    clear_gen(0);
    clear_gen(1);
    active_ = 0;
    occupied_ = 0;
    filter_bits_ = 8;
}

bool IcaoFilter::big_test(uint32_t addr) const {
    for (int g = 0; g < 2; ++g)
        if (std::find(big_[g].begin(), big_[g].end(), addr) != big_[g].end()) return true;
    return false;
}

void IcaoFilter::resize(uint32_t bits) {
    // two fresh tables; only the ACTIVE generation's entries are re-inserted
    filter_bits_ = bits;
    clear_gen(active_ ^ 1);
    occupied_ = (uint32_t) (members_[active_].size() + big_[active_].size());
}

void IcaoFilter::add(uint32_t addr) {
    bool inserted = false;
    if (addr < (1u << 24)) {
        uint64_t &w = bits_[active_][addr >> 6];
        const uint64_t bit = 1ull << (addr & 63);
        if (!(w & bit)) {
            w |= bit; members_[active_].push_back(addr); inserted = true;
            if (news_ && !((bits_[active_ ^ 1][addr >> 6] >> (addr & 63)) & 1)) news_->push_back(addr);
        }
    } else if (std::find(big_[active_].begin(), big_[active_].end(), addr) == big_[active_].end()) {
        big_[active_].push_back(addr);
        inserted = true;
    }
    if (inserted) ++occupied_;
    if (occupied_ > (1u << filter_bits_) / 3 && filter_bits_ < 20) resize(filter_bits_ + 1);
}

void IcaoFilter::expire() {
    if (occupied_ < (1u << filter_bits_) / 9 && filter_bits_ > 8) resize(filter_bits_ - 1);
    occupied_ = 0;
    clear_gen(active_ ^ 1);
    active_ ^= 1;
}

void IcaoFilter::snapshot(Snapshot &s) const {
    for (int g = 0; g < 2; ++g) { s.members[g] = members_[g]; s.big[g] = big_[g]; }
    s.active = active_; s.occupied = occupied_; s.filter_bits = filter_bits_;
}

void IcaoFilter::restore(const Snapshot &s) {
    std::vector<uint32_t> *keep = drops_;
    drops_ = nullptr;   // (news_ is only touched by add)
    for (int g = 0; g < 2; ++g) {
        clear_gen(g);
        for (uint32_t a : s.members[g]) bits_[g][a >> 6] |= 1ull << (a & 63);
        members_[g] = s.members[g];
        big_[g] = s.big[g];
    }
    active_ = s.active; occupied_ = s.occupied; filter_bits_ = s.filter_bits;
    drops_ = keep;
}

void IcaoFilter::union_sorted(std::vector<uint32_t> &out) const {
    out.clear();
    out.insert(out.end(), members_[0].begin(), members_[0].end());
    out.insert(out.end(), members_[1].begin(), members_[1].end());
    std::sort(out.begin(), out.end());
    out.erase(std::unique(out.begin(), out.end()), out.end());
}

bool IcaoFilter::same_as(const IcaoFilter &o) const {
    if (occupied_ != o.occupied_ || filter_bits_ != o.filter_bits_) return false;
    for (int g = 0; g < 2; ++g) {
        std::vector<uint32_t> x = members_[g ? active_ ^ 1 : active_], y = o.members_[g ? o.active_ ^ 1 : o.active_];
        std::vector<uint32_t> bx = big_[g ? active_ ^ 1 : active_], by = o.big_[g ? o.active_ ^ 1 : o.active_];
        std::sort(x.begin(), x.end()); std::sort(y.begin(), y.end());
        std::sort(bx.begin(), bx.end()); std::sort(by.begin(), by.end());
        if (x != y || bx != by) return false;
    }
    return true;
}

bool Resolver::apply_device_walk(const uint32_t *per_buf, const uint32_t *adds, uint32_t nbuf, int32_t flip) {
    IcaoFilter::Snapshot snap;
    filter_.snapshot(snap);
    const int64_t now0 = synthetic_now_, flip0 = next_flip_;
    const uint64_t nflips0 = nflips_;
    bool ok = true;
    int32_t flipped_at = 0x7fffffff;
    for (uint32_t b = 0; ok && b < nbuf; ++b) {
        const uint32_t *row = per_buf + 6 * (size_t) b;
        const uint32_t *a = adds + row[4];
        for (uint32_t i = 0; ok && i < row[1]; ++i) {
            const uint32_t bits = filter_.table_bits();
            filter_.add(a[i]);
            ok = filter_.table_bits() == bits;
        }
        synthetic_now_ = (int64_t) ((uint64_t) row[2] | ((uint64_t) row[3] << 32));
        const uint64_t f = nflips_;
        after_buffer();
        if (nflips_ != f) { if (flipped_at != 0x7fffffff) ok = false; flipped_at = (int32_t) b; }
    }
    if (flipped_at != flip) ok = false;
    if (!ok) {
        filter_.restore(snap);
        synthetic_now_ = now0; next_flip_ = flip0; nflips_ = nflips0;
    }
    return ok;
}

void Resolver::reset(int64_t startup_ms, int clock_mode) {
    filter_.init();
    // external clock: the host mirrors every icaoFilterAdd() it performs, modesInit's icaoFilterAdd(Modes.show_only) included — with
    // a user --show-only a default entry added here as well would leave `occupied` one ahead of the host's and the resize
    // thresholds (icao_filter.c:65-93) firing on different adds
    if (clock_mode != 2) filter_.add(kShowOnlyDefault);
    synthetic_now_ = startup_ms;   // Modes.synthetic_now armed by ifileOpen (sdr_ifile.c:131-133)
    next_flip_ = 0;                // static next_flip = 0 (readsb.c:1227)
    nflips_ = 0;
    // The decode thread's first backgroundTasks() call comes before or after its first buffer (readsb.c:857-902):
    // before = the first expiry hits the filter while it only holds show_only, the next is due 60 s after start-up.
    if (clock_mode == 1) after_buffer();
    // external clock: the host forwards its own icaoFilterExpire() calls; nothing is ever due here
    if (clock_mode == 2) next_flip_ = std::numeric_limits<int64_t>::max();
}

bool Resolver::scheduled(int64_t buffer_ts) const {
    const int64_t *e = std::lower_bound(sched_, sched_ + nsched_, buffer_ts);
    return e != sched_ + nsched_ && *e == buffer_ts;
}

void Resolver::after_buffer(int64_t buffer_ts) {
    // backgroundTasks(now = mstime()) after every buffer (readsb.c:899-902, 1227-1231)
    if (clock_log_) clock_log_->push_back(synthetic_now_);
    bool due = synthetic_now_ >= next_flip_;
    if (sched_ && buffer_ts != kNoBufferTs) {       // an imposed schedule (several ranks walk one capture): it decides, the rule is only compared
        const bool imposed = scheduled(buffer_ts);
        if (imposed != due) ++sched_mismatch_;
        due = imposed;
    }
    if (due) {
        exp_lo_ = chunk_drops_.size();         // (what the expiry takes out of the union, when the changes are being tracked)
        filter_.expire();
        exp_hi_ = chunk_drops_.size();
        next_flip_ = synthetic_now_ + kFilterTtlMs;
        ++nflips_;
    }
}

void Resolver::tick_empty(int64_t sysTimestamp, int64_t sampleTimestamp) {
    synthetic_now_ = sysTimestamp;   // demod_2400.c:283-285
    after_buffer(sampleTimestamp);
}

void Resolver::reset_empty(int64_t startup_ms) {
    filter_.init();
    synthetic_now_ = startup_ms;
    next_flip_ = 0;
    nflips_ = 0;
}

// state blob: magic | synthetic_now | next_flip | nflips | occupied, table bits | four counts | active members (< 2^24), inactive
// members, active big, inactive big — each list ascending
static constexpr uint64_t kStateMagic = 0x3154415453524c46ull;   // "FLRSTAT1"

void Resolver::export_state(std::vector<uint8_t> &blob) const {
    IcaoFilter::Snapshot sn;
    filter_.snapshot(sn);
    const int a = sn.active;
    std::vector<uint32_t> lists[4] = {sn.members[a], sn.members[a ^ 1], sn.big[a], sn.big[a ^ 1]};
    for (auto &l : lists) std::sort(l.begin(), l.end());
    const uint64_t head[5] = {kStateMagic, (uint64_t) synthetic_now_, (uint64_t) next_flip_, nflips_, (uint64_t) sn.occupied | ((uint64_t) sn.filter_bits << 32)};
    const uint32_t counts[4] = {(uint32_t) lists[0].size(), (uint32_t) lists[1].size(), (uint32_t) lists[2].size(), (uint32_t) lists[3].size()};
    blob.clear();
    auto put = [&](const void *p, size_t n) { blob.insert(blob.end(), (const uint8_t *) p, (const uint8_t *) p + n); };
    put(head, sizeof(head));
    put(counts, sizeof(counts));
    for (auto &l : lists) if (!l.empty()) put(l.data(), l.size() * sizeof(uint32_t));
    while (blob.size() % 8) blob.push_back(0);
}

bool Resolver::import_state(const uint8_t *blob, size_t bytes) {
    uint64_t head[5];
    uint32_t counts[4];
    if (!blob || bytes < sizeof(head) + sizeof(counts)) return false;
    std::memcpy(head, blob, sizeof(head));
    std::memcpy(counts, blob + sizeof(head), sizeof(counts));
    const uint64_t total = (uint64_t) counts[0] + counts[1] + counts[2] + counts[3];
    const uint32_t bits = (uint32_t) (head[4] >> 32);
    if (head[0] != kStateMagic || bits < 8 || bits > 20 || total > (bytes - sizeof(head) - sizeof(counts)) / sizeof(uint32_t)) return false;
    const uint8_t *p = blob + sizeof(head) + sizeof(counts);
    IcaoFilter::Snapshot sn;
    sn.active = 0;
    std::vector<uint32_t> *dst[4] = {&sn.members[0], &sn.members[1], &sn.big[0], &sn.big[1]};
    for (int i = 0; i < 4; ++i) {
        dst[i]->resize(counts[i]);
        if (counts[i]) std::memcpy(dst[i]->data(), p, (size_t) counts[i] * sizeof(uint32_t));
        p += (size_t) counts[i] * sizeof(uint32_t);
        for (uint32_t a : *dst[i]) if ((i < 2) != (a < (1u << 24))) return false;      // a list in the wrong place: not one of ours
    }
    sn.occupied = (uint32_t) head[4];
    sn.filter_bits = bits;
    filter_.restore(sn);
    synthetic_now_ = (int64_t) head[1]; next_flip_ = (int64_t) head[2]; nflips_ = head[3];
    return true;
}

static inline int frame_bits(const PhaseRec &r) { return (r.msg[0] & 0x80) ? 112 : 56; }   // demod_2400.c:399, DF as sliced

// The accept walk over buffers [b_lo, b_hi) of a chunk, starting at live record rec_lo.  `pol` answers
// icaoFilterTest, takes icaoFilterAdd and is told the clock (Modes.synthetic_now) at every buffer end.
// recs ends with a sentinel (pos = 0xFFFFFFFF).  Indices written are chunk-relative.
template <class Policy>
int64_t Resolver::walk_range(Policy &pol, const PhaseRec *recs, uint64_t rec_lo, const BufferClock *bufs, uint32_t b_lo, uint32_t b_hi,
                             Accepted *out, uint32_t *aux_pos, uint16_t *aux_skip, uint32_t *aux_limit, uint64_t aux_cap,
                             ResolveCounts &counts) const {
    ResolveCounts c;                                           // locals: the optimiser keeps them out of memory
    const PhaseRec *r = recs + rec_lo;
    uint64_t nout = 0;
    for (uint32_t bi = b_lo; bi < b_hi; ++bi) {
        const BufferClock &b = bufs[bi];
        int64_t now = b.sysTimestamp;                          // Modes.synthetic_now, demod_2400.c:283-285
        const uint64_t end = (uint64_t) b.first + b.length;
        int64_t skip_until = -1;                               // the skip never crosses a buffer (loop-local pa)
        while (r->pos < end) {
            const uint32_t pos = r->pos;
            if ((int64_t) pos <= skip_until) {                 // hidden by the previous frame (:468)
                uint32_t all_cond = REC_COND;
                do { all_cond &= r->flags; ++r; } while (r->pos == pos);
                if (all_cond) ++c.skipped_cond_groups; else ++c.skipped_uncond_groups;
                continue;
            }
            // best over the tried phases, in phase order, strict '>' (demod_2400.c:246)
            int best = -2;
            const PhaseRec *br = nullptr;
            bool best_known = false;
            uint32_t all_cond = REC_COND;
            do {
                all_cond &= r->flags;
                const bool fixed = r->score_known == r->score_unknown;
                const bool known = fixed || pol.test(r->addr);         // for fixed scores either answer scores the same
                const int s = known ? r->score_known : r->score_unknown;
                if (s > best) { best = s; br = r; best_known = known; }
                ++r;
            } while (r->pos == pos);
            ++c.visited_groups;
            if (all_cond) ++c.visited_cond_groups; else ++c.visited_uncond_groups;
            if (best < 0) {                                    // demod_2400.c:390-397
                if (best == -1) ++c.rejected_unknown; else ++c.rejected_bad;
                continue;
            }
            const int msglen = frame_bits(*br);
            const int64_t rel = (int64_t) (pos - b.first) * 5 + (8 + 56) * 12 + br->phase;   // timestamp - sampleTimestamp, :406
            now = b.sysTimestamp + rel / 12000;                                              // :409-414
            // decodeModesMessage's CRC/address stage: same filter state as the scoring above
            const bool accept = (br->flags & REC_ACCEPT_IF_UNKNOWN) || best_known;
            if (!accept) { ++c.rejected_unknown; continue; }   // :423-429, no skip-ahead
            if (br->flags & REC_ADDER) pol.add(br->addr & 0xffffffu);       // mode_s.c:766-779
            ++c.accepted[(br->flags >> REC_CORR_SHIFT) & 3];
            ++c.best_phase[br->phase - 4];
            if (nout >= aux_cap) { counts.add(c); return -1; }
            out[nout] = Accepted{(uint32_t) (br - recs), bi, best};
            aux_pos[nout] = pos;
            aux_skip[nout] = (uint16_t) (msglen * 8 / 4);     // :468
            aux_limit[nout] = (uint32_t) end;
            ++nout;
            skip_until = (int64_t) pos + msglen * 8 / 4;
        }
        pol.buffer_end(now, b.sampleTimestamp);
    }
    counts.add(c);
    return (int64_t) nout;
}

namespace {
// the true filter: icao_filter.c semantics, the 60 s clock after every buffer (readsb.c:1227-1231)
struct LivePolicy {
    IcaoFilter &flt;
    int64_t &synthetic_now;
    Resolver &res;
    void (Resolver::*tick)(int64_t);
    bool test(uint32_t a) const { return flt.test(a); }
    void add(uint32_t a) { flt.add(a); }
    void buffer_end(int64_t now, int64_t buffer_ts) { synthetic_now = now; (res.*tick)(buffer_ts); }
};

// speculation: membership as it was when the chunk started, plus the range's own adds
struct SpecPolicy {
    const IcaoFilter &flt;
    SegmentWalk &w;
    bool test(uint32_t a) {
        if (a >> 24) { w.odd = true; return flt.test(a); }
        if (w.added.test(a)) return true;
        w.q_pre.set(a);
        // Behind the batch's expiry the generation that was inactive when the batch started is gone and the one that was active is
        // the other, still known, one: exactly predictable, so a range does not have to be walked again because an expiry fell
        // into the chunk (round 3; until then every range behind it failed its commit on the first address the expiry had dropped).
        return (w.flipped ? flt.in_generation(a, true) : flt.test(a)) || w.assumed.test(a);
    }
    void add(uint32_t a) {
        w.added.set(a);
        if (!w.recorded.test(a)) { w.recorded.set(a); w.adds.push_back(a); }
    }
    // the 60 s expiry clock (readsb.c:1227-1231) with the threshold the batch started with: exact as long as
    // the decisions are, which is what commit_segment establishes
    void buffer_end(int64_t now, int64_t buffer_ts) {
        w.adds_end.push_back((uint32_t) w.adds.size());
        w.end_clock.push_back(now);
        if (w.sched ? buffer_ts == w.flip_ts : now >= w.flip_clock) {
            w.flip_at = (int32_t) (w.b_lo + w.end_clock.size() - 1);
            ++w.nflip;
            w.flip_clock = now + kFilterTtlMs;
            w.flipped = true;
            w.recorded.clear();          // after an expiry the active generation is empty: every address counts again
        }
    }
};

uint64_t first_record_at(const PhaseRec *recs, uint64_t nrecs, uint32_t pos) {
    uint64_t lo = 0, hi = nrecs;
    while (lo < hi) { const uint64_t mid = (lo + hi) / 2; if (recs[mid].pos < pos) lo = mid + 1; else hi = mid; }
    return lo;
}
}  // namespace

namespace {
// what one wave of k_walk knows: the generations as they were when the chunk started, the table of first adds, its own adds
struct ModelPolicy {
    const IcaoFilter &flt;
    const std::vector<std::pair<uint32_t, uint32_t>> &first_cur;   // (address, first buffer that adds it), sorted by address
    uint32_t b;
    int32_t flip;
    AddrSet &own;
    int64_t end_clock = 0;
    bool odd = false;
    bool test(uint32_t a) {
        if (a >> 24) { odd = true; return false; }
        if (flt.in_generation(a, true) || ((int32_t) b <= flip && flt.in_generation(a, false)) || own.test(a)) return true;
        auto it = std::lower_bound(first_cur.begin(), first_cur.end(), std::make_pair(a, 0u));
        return it != first_cur.end() && it->first == a && it->second < b;
    }
    void add(uint32_t a) { own.set(a); }
    void buffer_end(int64_t now, int64_t) { end_clock = now; }
};
}  // namespace

int64_t Resolver::device_walk_model(const PhaseRec *recs, uint64_t nrecs, const std::vector<BufferClock> &buffers, std::vector<Accepted> &acc,
                                    ResolveCounts &counts, uint32_t max_walks, uint32_t *walks) {
    const uint32_t nbuf = (uint32_t) buffers.size();
    const int32_t no_flip = 0x7fffffff;
    std::vector<std::pair<uint32_t, uint32_t>> cur, next;            // the two tables
    int32_t flip = no_flip;
    for (uint32_t b = 0; b < nbuf; ++b) if (buffers[b].sysTimestamp >= next_flip_) { flip = (int32_t) b; break; }   // k_walk_begin's guess
    std::vector<uint32_t> rows((size_t) 6 * nbuf), adds;
    std::vector<std::vector<uint32_t>> buf_adds(nbuf);
    std::vector<int64_t> end_clock(nbuf);
    std::vector<Accepted> out;
    std::vector<uint32_t> pos(nrecs + 1), lim(nrecs + 1);
    std::vector<uint16_t> skip(nrecs + 1);
    AddrSet own;
    own.ensure();
    ResolveCounts rc;
    bool settled = false;
    uint32_t w = 0;
    while (w < max_walks && !settled) {
        ++w;
        out.assign(nrecs + 1, Accepted{});
        rc = ResolveCounts();
        next.clear();
        uint64_t nout = 0;
        bool odd = false;
        for (uint32_t b = 0; b < nbuf; ++b) {                        // (on the device: all at once, a wave each)
            own.clear();
            ModelPolicy pol{filter_, cur, b, flip, own};
            const uint64_t lo = first_record_at(recs, nrecs, buffers[b].first);
            const int64_t n = walk_range(pol, recs, lo, buffers.data(), b, b + 1, out.data() + nout, pos.data(), skip.data(), lim.data(), nrecs + 1 - nout, rc);
            if (n < 0 || pol.odd) { odd = true; break; }
            rows[6 * b + 0] = (uint32_t) n; rows[6 * b + 5] = (uint32_t) nout;
            nout += (uint64_t) n;
            end_clock[b] = pol.end_clock;
            buf_adds[b] = own.list;
            for (uint32_t a : own.list)                              // only what is not known for the whole chunk anyway
                if (!(filter_.in_generation(a, true) || (flip == no_flip && filter_.in_generation(a, false)))) next.emplace_back(a, b);
        }
        if (odd) { if (walks) *walks = w; return -1; }
        std::sort(next.begin(), next.end());                         // first add per address = the smallest buffer
        next.erase(std::unique(next.begin(), next.end(), [](const std::pair<uint32_t, uint32_t> &x, const std::pair<uint32_t, uint32_t> &y) { return x.first == y.first; }), next.end());
        int32_t f = no_flip;
        for (uint32_t b = 0; b < nbuf; ++b) if (end_clock[b] >= next_flip_) { f = (int32_t) b; break; }
        settled = next == cur && f == flip;
        if (!settled) { cur.swap(next); flip = f; }
        else acc.assign(out.begin(), out.begin() + (std::ptrdiff_t) nout);
    }
    if (walks) *walks = w;
    if (!settled) return -1;
    adds.clear();
    for (uint32_t b = 0; b < nbuf; ++b) {
        rows[6 * b + 1] = (uint32_t) buf_adds[b].size();
        rows[6 * b + 2] = (uint32_t) (uint64_t) end_clock[b]; rows[6 * b + 3] = (uint32_t) ((uint64_t) end_clock[b] >> 32);
        rows[6 * b + 4] = (uint32_t) adds.size();
        adds.insert(adds.end(), buf_adds[b].begin(), buf_adds[b].end());
    }
    adds.push_back(0);                                               // (never empty: data() stays valid)
    if (!apply_device_walk(rows.data(), adds.data(), nbuf, flip)) return -1;
    counts.add(rc);
    return (int64_t) acc.size();
}

int64_t Resolver::decide(const PhaseRec *recs, uint64_t nrecs, const std::vector<BufferClock> &buffers, std::vector<Accepted> &acc,
                         uint32_t *aux_pos, uint16_t *aux_skip, uint32_t *aux_limit, uint64_t aux_cap, ResolveCounts &counts) {
    (void) nrecs;
    if (acc.size() < aux_cap) acc.resize(aux_cap);             // a pre-sized array, its length is the return value
    LivePolicy pol{filter_, synthetic_now_, *this, &Resolver::after_buffer};
    return walk_range(pol, recs, 0, buffers.data(), 0, (uint32_t) buffers.size(), acc.data(), aux_pos, aux_skip, aux_limit, aux_cap, counts);
}

void Resolver::collect_adders(const PhaseRec *recs, SegmentWalk &w) const {
    w.cand_seen.ensure();
    w.cand_seen.clear();
    w.candidates.clear();
    for (uint64_t i = w.rec_lo; i < w.rec_hi; ++i) {
        if (!(recs[i].flags & REC_ADDER)) continue;
        const uint32_t a = recs[i].addr & 0xffffffu;
        // not in the ACTIVE generation: a new address, or one the expiry would drop had this frame not refreshed it (a range behind
        // the expiry takes "known" from the active generation + these)
        if (filter_.in_generation(a, true) || w.cand_seen.test(a)) continue;
        w.cand_seen.set(a);
        w.candidates.push_back(a);
    }
}

void Resolver::spec_walk(const PhaseRec *recs, const std::vector<BufferClock> &buffers, SegmentWalk &w) const {
    w.added.ensure(); w.q_pre.ensure(); w.recorded.ensure();
    w.added.clear(); w.q_pre.clear(); w.recorded.clear();
    w.flip_at = -1; w.nflip = 0;
    w.flipped = w.after_flip;
    w.adds.clear(); w.adds_end.clear(); w.end_clock.clear();
    w.counts = ResolveCounts();
    w.speculated = false;
    // a range cannot accept more frames than it has records
    const uint64_t cap = w.rec_hi - w.rec_lo + 1;
    if (w.acc.size() < cap) { w.acc.resize(cap); w.pos.resize(cap); w.limit.resize(cap); w.skip.resize(cap); }
    SpecPolicy pol{filter_, w};
    const int64_t n = walk_range(pol, recs, w.rec_lo, buffers.data(), w.b_lo, w.b_hi, w.acc.data(), w.pos.data(), w.skip.data(),
                                 w.limit.data(), cap, w.counts);
    if (n < 0) { w.odd = true; w.nacc = 0; } else w.nacc = (uint64_t) n;
}

bool Resolver::commit_segment(const PhaseRec *recs, const std::vector<BufferClock> &buffers, SegmentWalk &w) {
    (void) recs;
    bool ok = !w.odd;
    // (1) what the range assumed about the state at its start against the true filter: adds it expected from
    //     earlier ranges that did not happen, and anything the earlier ranges of the batch dropped (or dropped
    //     and brought back: the trackers only say "changed")
    for (size_t i = 0; ok && i < w.assumed.list.size(); ++i) {
        const uint32_t a = w.assumed.list[i];
        if (!filter_.test(a) && w.q_pre.test(a)) ok = false;
    }
    // (what the batch's expiry dropped is what a range behind it left out of "known" itself)
    for (size_t i = 0; ok && i < chunk_drops_.size(); ++i) {
        if (w.after_flip && i >= exp_lo_ && i < exp_hi_) continue;
        ok = !w.q_pre.test(chunk_drops_[i]);
    }
    for (size_t i = 0; ok && i < chunk_news_.size(); ++i) {
        const uint32_t a = chunk_news_[i];
        if (!w.assumed.test(a) && w.q_pre.test(a)) ok = false;
    }
    // (2) catch the true filter up: the range's first adds in order, the clock after every buffer.  Whatever an
    //     expiry or a resize inside the range drops must not be something the range asked about; the expiry must
    //     come where the range itself put it (it restarted its list of first adds there), and only once (a
    //     second one could drop the range's own adds).
    const size_t drops0 = chunk_drops_.size(), news0 = chunk_news_.size(), exp_lo0 = exp_lo_, exp_hi0 = exp_hi_;
    const int64_t now0 = synthetic_now_, flip0 = next_flip_;
    const uint64_t nflips0 = nflips_, mism0 = sched_mismatch_;
    const size_t log0 = clock_log_ ? clock_log_->size() : 0;
    if (ok) ok = w.nflip <= 1;
    if (ok) {
        IcaoFilter::Snapshot snap;
        filter_.snapshot(snap);
        uint32_t k = 0;
        int32_t flipped_at = -1;
        for (uint32_t bi = w.b_lo; ok && bi < w.b_hi; ++bi) {
            const uint32_t e = w.adds_end[bi - w.b_lo];
            for (; k < e; ++k) filter_.add(w.adds[k]);
            synthetic_now_ = w.end_clock[bi - w.b_lo];
            const uint64_t f = nflips_;
            after_buffer(buffers[bi].sampleTimestamp);
            if (nflips_ != f) { if (flipped_at >= 0) ok = false; flipped_at = (int32_t) bi; }
        }
        if (flipped_at != w.flip_at) ok = false;
        // the range's own expiry, where it put it itself: it answered from both generations before it and from the surviving one
        // behind it — what the expiry drops is accounted for; what a resize drops is not
        for (size_t i = drops0; ok && i < chunk_drops_.size(); ++i) {
            if (flipped_at >= 0 && i >= exp_lo_ && i < exp_hi_) {
                // ... except an address the range took for refreshed by an earlier range's frame (assumed) that was not: before
                // its expiry both answers are "known", behind it only the assumption's
                if (w.assumed.test(chunk_drops_[i]) && w.q_pre.test(chunk_drops_[i])) ok = false;
                continue;
            }
            // (... nor is the range's own add of an address before its expiry, which a growing table behind the expiry drops
            // with the rest of that generation: the range would go on answering "known" from its own set)
            ok = !w.q_pre.test(chunk_drops_[i]) && !w.added.test(chunk_drops_[i]);
        }
        if (!ok) {
            filter_.restore(snap);
            chunk_drops_.resize(drops0); chunk_news_.resize(news0); exp_lo_ = exp_lo0; exp_hi_ = exp_hi0;
            synthetic_now_ = now0; next_flip_ = flip0; nflips_ = nflips0; sched_mismatch_ = mism0;
            if (clock_log_) clock_log_->resize(log0);
        }
    }
    w.speculated = ok;
    return ok;
}

void Resolver::serial_segment(const PhaseRec *recs, const std::vector<BufferClock> &buffers, SegmentWalk &w) {
    const uint64_t cap = w.rec_hi - w.rec_lo + 1;
    if (w.acc.size() < cap) { w.acc.resize(cap); w.pos.resize(cap); w.limit.resize(cap); w.skip.resize(cap); }
    w.counts = ResolveCounts();
    w.speculated = false;
    LivePolicy pol{filter_, synthetic_now_, *this, &Resolver::after_buffer};
    const int64_t n = walk_range(pol, recs, w.rec_lo, buffers.data(), w.b_lo, w.b_hi, w.acc.data(), w.pos.data(), w.skip.data(),
                                 w.limit.data(), cap, w.counts);
    w.nacc = n < 0 ? 0 : (uint64_t) n;      // n < 0 cannot happen: a range accepts at most one frame per record
}

void Resolver::copy_state_from(const Resolver &o) {
    IcaoFilter::Snapshot snap;
    o.filter_.snapshot(snap);
    filter_.restore(snap);
    synthetic_now_ = o.synthetic_now_; next_flip_ = o.next_flip_; nflips_ = o.nflips_;
    sched_ = o.sched_; nsched_ = o.nsched_; sched_mismatch_ = o.sched_mismatch_; clock_log_ = o.clock_log_;
}

void Resolver::adopt(Resolver &shadow) {
    std::swap(filter_, shadow.filter_);        // vectors change hands, nothing is copied
    filter_.track_changes(nullptr, nullptr);
    shadow.filter_.track_changes(nullptr, nullptr);
    synthetic_now_ = shadow.synthetic_now_; next_flip_ = shadow.next_flip_; nflips_ = shadow.nflips_; sched_mismatch_ = shadow.sched_mismatch_;
    chunk_drops_.swap(shadow.chunk_drops_);
    chunk_news_.swap(shadow.chunk_news_);
    exp_lo_ = shadow.exp_lo_; exp_hi_ = shadow.exp_hi_;
}

void Resolver::parallel_walk(const PhaseRec *recs, uint64_t nrecs, const std::vector<BufferClock> &buffers, std::vector<SegmentWalk> &segs,
                             const Runner &run, uint64_t *batches) {
    (void) nrecs;
    const int K = (int) segs.size();
    if (!shadow_) shadow_.reset(new Resolver());
    int t0 = 0;
    while (t0 < K) {
        // one batch: ranges [t0, K) against the filter as it stands now
        const int nb = K - t0;
        if (batches) ++*batches;
        run(nb, [&](int i) { collect_adders(recs, segs[t0 + i]); });
        exp_lo_ = exp_hi_ = 0;
        for (int t = t0 + 1; t < K; ++t) {                   // assumed(t) = candidates of ranges t0 .. t-1
            SegmentWalk &w = segs[t];
            w.assumed.ensure();
            w.assumed.clear();
            for (int u = t0; u < t; ++u)
                for (uint32_t a : segs[u].candidates) w.assumed.set(a);
            // The expiry clock.  A range can tell from the buffer grid alone whether the expiry that is due has
            // happened before it starts (the clock at a buffer's end lies within the buffer's own 55 ms): surely
            // not -> it watches for it itself; surely yes -> the next one is 60 s away; the buffer in between, or
            // a chunk longer than the filter's TTL -> no speculation.
            w.odd = false;
            w.after_flip = false;
            w.sched = sched_ != nullptr;
            if (w.sched) {
                // an imposed schedule says exactly where the expiry falls: in an earlier range of the batch, in this one, or nowhere near
                const int64_t *lo = std::lower_bound(sched_, sched_ + nsched_, buffers[segs[t0].b_lo].sampleTimestamp);
                const int64_t *mid = std::lower_bound(lo, sched_ + nsched_, buffers[w.b_lo].sampleTimestamp);
                const int64_t *hi = std::upper_bound(mid, sched_ + nsched_, buffers[w.b_hi - 1].sampleTimestamp);
                w.flip_ts = kNoBufferTs;
                w.flip_clock = std::numeric_limits<int64_t>::max();
                if ((mid - lo) + (hi - mid) > 1) w.odd = true;         // two expiries inside one chunk: a chunk longer than the filter's TTL
                else if (mid - lo == 1) w.after_flip = true;
                else if (hi - mid == 1) w.flip_ts = *mid;
                continue;
            }
            const int64_t prev = buffers[w.b_lo - 1].sysTimestamp;
            if (prev + kBufferSpanMs < next_flip_) w.flip_clock = next_flip_;
            else if (prev >= next_flip_ && buffers[w.b_hi - 1].sysTimestamp + kBufferSpanMs < next_flip_ + kFilterTtlMs - kBufferSpanMs) {
                w.flip_clock = std::numeric_limits<int64_t>::max();
                w.after_flip = true;
            } else w.odd = true;
        }
        shadow_->copy_state_from(*this);
        shadow_->chunk_drops_.clear();
        shadow_->chunk_news_.clear();
        shadow_->exp_lo_ = shadow_->exp_hi_ = 0;
        shadow_->filter_.track_changes(&shadow_->chunk_drops_, &shadow_->chunk_news_);
        run(nb, [&](int i) {
            if (i == 0) shadow_->serial_segment(recs, buffers, segs[t0]);      // the true walk, on the private copy
            else spec_walk(recs, buffers, segs[t0 + i]);
        });
        adopt(*shadow_);                                      // range t0 is done; its drops / news are the trackers' start
        segs[t0].speculated = true;
        filter_.track_changes(&chunk_drops_, &chunk_news_);
        int t = t0 + 1;
        while (t < K && commit_segment(recs, buffers, segs[t])) ++t;
        filter_.track_changes(nullptr, nullptr);
        t0 = t;
    }
}

uint64_t segment_first_record(const PhaseRec *recs, uint64_t nrecs, uint32_t pos) { return first_record_at(recs, nrecs, pos); }

void flip_schedule(const int64_t *end_clock, uint64_t nbuf, int64_t startup_ms, int clock_mode, std::vector<uint64_t> &flip_after) {
    // Resolver::reset + after_buffer, nothing else: next_flip = 0 (static, readsb.c:1227); clock mode 1 = one expiry before buffer 0
    int64_t next_flip = clock_mode == 1 ? startup_ms + kFilterTtlMs : 0;
    flip_after.clear();
    for (uint64_t b = 0; b < nbuf; ++b)
        if (end_clock[b] >= next_flip) { flip_after.push_back(b); next_flip = end_clock[b] + kFilterTtlMs; }
}

uint64_t expiry_windows(uint64_t nbuf_total, uint32_t buf_samples, int64_t startup_ms, int clock_mode, uint8_t *mask) {
    if (nbuf_total == 0) return 0;
    std::memset(mask, 0, (size_t) nbuf_total);
    auto sys_ms = [&](uint64_t b) { return (int64_t) ((b * (uint64_t) buf_samples * 5) / 12000) + startup_ms; };
    // the first buffer whose clock start is >= t (the starts ascend)
    auto lower = [&](int64_t t) { uint64_t lo = 0, hi = nbuf_total; while (lo < hi) { const uint64_t mid = (lo + hi) / 2; if (sys_ms(mid) < t) lo = mid + 1; else hi = mid; } return lo; };
    const int64_t span = 56;
    int64_t lo, hi;
    if (clock_mode == 1) lo = hi = startup_ms + kFilterTtlMs;          // one expiry before buffer 0, the next due 60 s after start-up
    else { mask[0] = 1; lo = sys_ms(0) + kFilterTtlMs; hi = sys_ms(0) + span + kFilterTtlMs; }   // next_flip = 0: the first expiry follows buffer 0
    const int64_t end = sys_ms(nbuf_total - 1) + span;
    while (lo <= end) {
        uint64_t a = lower(lo - span), b = lower(hi + 1);              // buffers whose [start, start + 55] reaches lo and starts by hi
        a = a > 0 ? a - 1 : 0;                                          // (a buffer to spare on either side)
        b = b + 1 < nbuf_total ? b + 1 : nbuf_total;
        for (uint64_t i = a; i < b; ++i) mask[i] = 1;
        lo += kFilterTtlMs; hi += kFilterTtlMs + 111;
    }
    uint64_t n = 0;
    for (uint64_t i = 0; i < nbuf_total; ++i) n += mask[i];
    return n;
}

void estimate_end_clocks(const PhaseRec *recs, uint64_t nrecs, const std::vector<BufferClock> &buffers, std::vector<int64_t> &out) {
    uint64_t i = 0;
    for (const BufferClock &b : buffers) {
        int64_t now = b.sysTimestamp;
        const uint64_t bend = (uint64_t) b.first + b.length;
        while (i < nrecs && recs[i].pos < bend) {
            const uint32_t pos = recs[i].pos;
            int best = -2;
            uint32_t phase = 0;
            bool uncond = false;
            for (; i < nrecs && recs[i].pos == pos; ++i) {
                const PhaseRec &r = recs[i];
                if (!(r.flags & REC_COND)) uncond = true;
                if (r.score_known > best) { best = r.score_known; phase = r.phase; }
            }
            if (uncond && best >= 0) now = b.sysTimestamp + ((int64_t) (pos - b.first) * 5 + (8 + 56) * 12 + phase) / 12000;
        }
        out.push_back(now);
    }
}

int shard_walk_core(Resolver &res, const ShardWalkPlan &plan, size_t nchunks, const std::function<void(size_t, uint64_t &, uint64_t &)> &chunk,
                    const std::function<int(size_t, bool)> &walk, ShardWalkOut &out, const char **err) {
    static const char *none = "";
    *err = none;
    out.clocks.clear(); out.state_first.clear(); out.state_end.clear();
    for (size_t i = 1; i < plan.nsched; ++i)
        if (plan.sched[i] <= plan.sched[i - 1]) { *err = "the expiry schedule must be ascending"; return -1; }
    res.log_end_clocks(nullptr);
    bool own_started = false, have_state = false;
    uint64_t expect = 0;
    for (size_t i = 0; i < nchunks; ++i) {
        uint64_t pos = 0, n = 0;
        chunk(i, pos, n);
        if (i == 0) {
            if (pos > plan.own_first || pos % plan.buf_samples) { *err = "the chunks start behind the range's first sample"; return -1; }
            if (plan.start_state) {
                if (!res.import_state(plan.start_state, plan.start_state_bytes)) { *err = "not a filter state"; return -1; }
                have_state = true;
            } else if (pos == 0) res.reset(plan.startup_ms, plan.clock_mode);    // the stream's own start: the reference's initial state
            else res.reset_empty(plan.startup_ms);
            res.set_schedule(plan.sched, plan.nsched);
        } else if (pos != expect) { *err = "the chunks must be consecutive"; return -1; }
        expect = pos + n;
        if (pos < plan.own_first) {                            // warm-up: walked for the filter's state only
            if (expect > plan.own_first) { *err = "a chunk straddles the range's first sample (feed warm-up and range separately)"; return -1; }
            if (have_state) continue;                          // ... or not at all: the state at the range's first sample was given
            const int rc = walk(i, false);
            if (rc != 0) return rc;
            continue;
        }
        if (!own_started) {
            own_started = true;
            if (pos != plan.own_first) { *err = "no chunk starts at the range's first sample"; return -1; }
            if (!have_state) {
                // the expiries before the range, counted from the schedule (a cold start counted only those of its warm-up)
                const int64_t ts0 = (int64_t) plan.own_first * 5;
                res.set_nflips((uint64_t) (std::lower_bound(plan.sched, plan.sched + plan.nsched, ts0) - plan.sched) + (plan.clock_mode == 1 ? 1u : 0u));
            }
            res.export_state(out.state_first);
            res.log_end_clocks(&out.clocks);
        }
        const int rc = walk(i, true);
        if (rc != 0) { res.log_end_clocks(nullptr); return rc; }
    }
    res.log_end_clocks(nullptr);
    if (!own_started) { *err = "no chunk of the range itself"; return -1; }
    res.export_state(out.state_end);
    return 0;
}

void Resolver::build_messages(const PhaseRec *recs, const unsigned long long *sig, const unsigned long long *msig, const std::vector<BufferClock> &buffers,
                              const Accepted *acc, uint64_t nacc, mgpu_msg *out) {
    static_assert(sizeof(PhaseRec) == 32 && offsetof(PhaseRec, msg) == 16, "frame bytes are the record's second half");
    for (uint64_t n = 0; n < nacc; ++n) {
        const PhaseRec &r = recs[acc[n].rec];
        const BufferClock &b = buffers[acc[n].buffer];
        // the frame as two little-endian words: byte k of the frame = bits 8k.. of lo (k < 8) / hi (k >= 8)
        uint64_t lo, hi;
        std::memcpy(&lo, r.msg, 8);
        std::memcpy(&hi, r.msg + 8, 8);
        if (r.flags & REC_LONG) hi &= 0x0000ffffffffffffull; else { lo &= 0x00ffffffffffffffull; hi = 0; }
        uint64_t mlo = lo, mhi = hi;                           // msg = raw with the CRC repair applied
        if (r.flags & REC_DFFIX) mlo = (mlo & ~0xf8ull) | (17u << 3);      // fixDF17msgtype, mode_s.c:276-301
        else {
            if (r.fixbit0 != 0xff) { const uint64_t f = 0x80ull >> (r.fixbit0 & 7); if (r.fixbit0 < 64) mlo ^= f << (r.fixbit0 & 56); else mhi ^= f << (r.fixbit0 & 56); }
            if (r.fixbit1 != 0xff) { const uint64_t f = 0x80ull >> (r.fixbit1 & 7); if (r.fixbit1 < 64) mlo ^= f << (r.fixbit1 & 56); else mhi ^= f << (r.fixbit1 & 56); }
        }
        const unsigned msgtype = (unsigned) (mlo & 0xff) >> 3;
        const unsigned msgbits = (msgtype & 0x10) ? 112 : 56;
        if (msgbits == 56) { mlo &= 0x00ffffffffffffffull; mhi = 0; lo &= 0x00ffffffffffffffull; hi = 0; }
        mgpu_msg m;
        m.timestamp = b.sampleTimestamp + (int64_t) (r.pos - b.first) * 5 + (8 + 56) * 12 + r.phase;   // demod_2400.c:406
        m.sysTimestamp = b.sysTimestamp + (m.timestamp - b.sampleTimestamp) / 12000;                    // :409
        m.sig_sumsq = msig ? msig[n] & ~(1ull << 63) : sig ? sig[acc[n].rec] : 0;   // :442-445, from the GPU: per accepted frame (k_msg_sig), per live record, or filled in by the caller
        m.sig_len = (uint16_t) (frame_bits(r) * 12 / 5);       // :439
        m.score = (int16_t) acc[n].score;
        m.phase = r.phase;
        m.correctedbits = (r.flags >> REC_CORR_SHIFT) & 3;
        m.msgtype = (uint8_t) msgtype;
        m.msgbits = (uint8_t) msgbits;
        m.addr = r.addr & 0xffffffu;
        std::memcpy(m.msg, &mlo, 8); std::memcpy(m.msg + 8, &mhi, 6);
        std::memcpy(m.raw, &lo, 8); std::memcpy(m.raw + 8, &hi, 6);
#if defined(__SSE2__)
        if ((reinterpret_cast<uintptr_t>(out) & 15) == 0) {   // streaming stores: the consumer is another core, skip the RFO
            __m128i v[4];
            std::memcpy(v, &m, 64);
            __m128i *d = reinterpret_cast<__m128i *>(out + n);
            _mm_stream_si128(d + 0, v[0]); _mm_stream_si128(d + 1, v[1]);
            _mm_stream_si128(d + 2, v[2]); _mm_stream_si128(d + 3, v[3]);
        } else
#endif
        out[n] = m;
    }
#if defined(__SSE2__)
    _mm_sfence();
#endif
}

int64_t Resolver::walk(const PhaseRec *recs, const unsigned long long *sig, uint64_t nrecs,
                       const std::vector<BufferClock> &buffers, std::vector<mgpu_msg> &out, uint32_t *aux_pos,
                       uint16_t *aux_skip, uint32_t *aux_limit, uint64_t aux_cap, ResolveCounts &c) {
    std::vector<Accepted> acc;
    const int64_t n = decide(recs, nrecs, buffers, acc, aux_pos, aux_skip, aux_limit, aux_cap, c);
    if (n <= 0) return n;
    const size_t first = out.size();
    out.resize(first + (size_t) n);
    build_messages(recs, sig, nullptr, buffers, acc.data(), (uint64_t) n, out.data() + first);
    return n;
}

}  // namespace mgpu
