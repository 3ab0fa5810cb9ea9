// resolve.h — the ordered half of demodulate2400: best-phase selection, acceptance, skip-ahead
// and the ICAO address filter (demod_2400.c:381-472, mode_s.c:443-606/766-779, icao_filter.c).
//
// The kernels emit, for every candidate position and tried phase, a record whose score is known
// up to one question: "is this address in the ICAO filter right now?".  Answering it needs the
// exact serial order of the reference (accepted frames add addresses, accepted frames hide the
// next 112/224 positions, the filter forgets on a 60 s clock), so this walk is sequential per
// stream.  It touches only the records that survived the pre-screen (about one per real frame).
#pragma once
#include <cstdint>
#include <functional>
#include <memory>
#include <string>
#include <vector>

#include "../../include/modes_gpu.h"
#include "kernels.h"

namespace mgpu {

// Exact model of icao_filter.c: two generations, `occupied`, table size.  Bucket placement never
// shows in results; membership, the occupied count and the grow/shrink thresholds do, because
// growing re-inserts only the active generation (icaoFilterResize, icao_filter.c:65-93).
class IcaoFilter {
  public:
    IcaoFilter();
    void init();                       // icaoFilterInit (:47-59)
    void add(uint32_t addr);           // icaoFilterAdd (:112-130)
    bool test(uint32_t addr) const {   // icaoFilterTest (:132-154)
        return addr < (1u << 24) ? (((bits_[0][addr >> 6] | bits_[1][addr >> 6]) >> (addr & 63)) & 1) : big_test(addr);
    }
    void expire();                     // icaoFilterExpire (:96-110)
    uint32_t occupied() const { return occupied_; }
    uint32_t table_bits() const { return filter_bits_; }

    // --- support for the speculative parallel walk (resolve.cpp: Resolver::spec_walk / commit_segment) ---
    struct Snapshot {
        std::vector<uint32_t> members[2], big[2];
        int active = 0;
        uint32_t occupied = 0, filter_bits = 8;
    };
    void snapshot(Snapshot &s) const;
    void restore(const Snapshot &s);
    int active_index() const { return active_; }
    void union_sorted(std::vector<uint32_t> &out) const;      // addresses < 2^24 in either generation, ascending
    const std::vector<uint32_t> &members(bool active) const { return members_[active ? active_ : active_ ^ 1]; }   // addresses < 2^24
    bool same_as(const IcaoFilter &o) const;                  // membership per generation, occupied, table size
    bool in_generation(uint32_t addr, bool active) const {    // addr < 2^24
        return (bits_[active ? active_ : active_ ^ 1][addr >> 6] >> (addr & 63)) & 1;
    }
    // addresses that leave the union (expire / resize) and addresses that enter it (first add)
    void track_changes(std::vector<uint32_t> *drops, std::vector<uint32_t> *news) { drops_ = drops; news_ = news; }

  private:
    void resize(uint32_t bits);
    bool big_test(uint32_t addr) const;
    void clear_gen(int g);
    std::vector<uint64_t> bits_[2];       // 2^24-bit membership bitmap per generation
    std::vector<uint32_t> members_[2];    // for O(members) clearing
    std::vector<uint32_t> big_[2];        // addresses >= 2^24 (only Modes.show_only's default)
    int active_ = 0;
    uint32_t occupied_ = 0, filter_bits_ = 8;
    std::vector<uint32_t> *drops_ = nullptr, *news_ = nullptr;
};

struct BufferClock {
    int64_t sampleTimestamp;   // mag_buf.sampleTimestamp (12 MHz ticks)
    int64_t sysTimestamp;      // mag_buf.sysTimestamp (ms)
    uint32_t first, length;    // scan positions [first, first+length) of the feed
};

struct ResolveCounts {
    void add(const ResolveCounts &o) {
        visited_groups += o.visited_groups; rejected_unknown += o.rejected_unknown; rejected_bad += o.rejected_bad;
        for (int i = 0; i < 3; ++i) accepted[i] += o.accepted[i];
        for (int i = 0; i < 5; ++i) best_phase[i] += o.best_phase[i];
        skipped_uncond_groups += o.skipped_uncond_groups; skipped_cond_groups += o.skipped_cond_groups;
        visited_cond_groups += o.visited_cond_groups; visited_uncond_groups += o.visited_uncond_groups;
    }
    uint64_t visited_groups = 0;        // candidate groups the walk looked at
    uint64_t rejected_unknown = 0;      // visited, best == -1 or decode stage -1
    uint64_t rejected_bad = 0;          // visited, decode stage -2 (not produced) / best == -2
    uint64_t accepted[3] = {0, 0, 0};
    uint64_t best_phase[5] = {0, 0, 0, 0, 0};
    uint64_t skipped_uncond_groups = 0; // live groups hidden by a skip window, >= 1 unconditional record
    uint64_t skipped_cond_groups = 0;   // live groups hidden by a skip window, conditional records only
    uint64_t visited_cond_groups = 0;   // visited groups with conditional records only
    uint64_t visited_uncond_groups = 0;
};

// One accepted frame, as the ordered walk decided it: which live record won, in which buffer, with
// what score.  Everything else about the message is a pure function of these (build_messages).
struct Accepted {
    uint32_t rec;      // index into the chunk's live records
    uint32_t buffer;   // index into the chunk's buffer grid
    int32_t score;
};

// A sparse set over 24-bit addresses: bitmap + list of what is set (for O(set) clearing).
struct AddrSet {
    std::vector<uint64_t> bits;
    std::vector<uint32_t> list;
    bool test(uint32_t a) const { return (bits[a >> 6] >> (a & 63)) & 1; }
    void set(uint32_t a) {
        uint64_t &w = bits[a >> 6];
        const uint64_t b = 1ull << (a & 63);
        if (!(w & b)) { w |= b; list.push_back(a); }
    }
    void clear() { for (uint32_t a : list) bits[a >> 6] = 0; list.clear(); }
    void ensure() { if (bits.empty()) bits.assign(1u << 18, 0); }
};

// One contiguous range of buffers of a chunk, walked speculatively by one thread against the
// filter as it stood at the start of the chunk (Resolver::spec_walk), then validated and folded
// into the true filter state in stream order (Resolver::commit_segment).
struct SegmentWalk {
    // the range
    uint32_t b_lo = 0, b_hi = 0;          // buffers [b_lo, b_hi) of the chunk
    uint64_t rec_lo = 0, rec_hi = 0;      // live records [rec_lo, rec_hi) = those at positions inside the range
    // decisions (the same outputs as Resolver::decide, indices chunk-relative)
    std::vector<Accepted> acc;
    std::vector<uint32_t> pos, limit;
    std::vector<uint16_t> skip;
    uint64_t nacc = 0;
    ResolveCounts counts;
    // what the true filter needs to catch up: per buffer, the first icaoFilterAdd of every address since the last
    // expiry (repeats are no-ops), and the clock at each buffer end
    std::vector<uint32_t> adds, adds_end;
    std::vector<int64_t> end_clock;
    // what the decisions assumed
    AddrSet added;                        // own adds so far
    AddrSet recorded;                     // own adds since the range's last expiry (what `adds` leaves out as repeats)
    bool flipped = false;                 // the expiry lies behind: known = the batch's ACTIVE generation (+ adds), not both
    bool after_flip = false;              // ... from the range's first buffer on (the expiry falls into an earlier range)
    int64_t flip_clock = 0;               // the range expires the filter at the first buffer end with clock >= this
    bool sched = false;                   // an imposed expiry schedule (Resolver::set_schedule): the range expires the filter after the
    int64_t flip_ts = 0;                  // buffer whose sampleTimestamp is flip_ts (INT64_MIN: after none), whatever the clock says
    int32_t flip_at = -1;                 // buffer after which it did (-1: never), nflip = how often
    int32_t nflip = 0;
    AddrSet q_pre;                        // addresses asked about before their own first add in this range
    AddrSet assumed;                      // addresses assumed to have been added by the earlier ranges of the batch
    AddrSet cand_seen;
    std::vector<uint32_t> candidates;     // addresses of this range's adder records the filter does not hold yet
    bool odd = false;                     // something the speculation does not model (address >= 2^24, overflow)
    bool speculated = false;              // outcome: false = re-walked serially by commit_segment
};

class Resolver {
  public:
    // clock_mode = mgpu_config.filter_clock: 0 first expiry after buffer 0, 1 before it, 2 never (external_expire only)
    void reset(int64_t startup_ms, int clock_mode = 0);
    // --- one capture walked by several ranks (config 5; api.cpp: mgpu_shard_walk, readsb_amd/shard.py) ---
    // A rank that starts in the middle of the stream: nothing known (not even modesInit's show_only entry, which is two expiries
    // gone by then), the clock wherever the first buffer puts it.
    void reset_empty(int64_t startup_ms);
    // An imposed expiry schedule: the filter expires after exactly the buffers whose sampleTimestamp is listed (ascending; the
    // array must outlive its use; nullptr = the reference's own rule, readsb.c:1227-1231).  The clock state (next_flip) is still
    // kept as the reference keeps it, so a schedule that is the fixed point of the walk's own end clocks reproduces the
    // reference's expiries exactly; mismatches() counts the buffers where the rule and the schedule disagreed.
    void set_schedule(const int64_t *buffer_ts, size_t n) { sched_ = buffer_ts; nsched_ = n; sched_mismatch_ = 0; }
    uint64_t schedule_mismatches() const { return sched_mismatch_; }
    void set_nflips(uint64_t n) { nflips_ = n; }
    // every buffer's end clock (Modes.synthetic_now when backgroundTasks looks at it), in stream order, appended to *log
    void log_end_clocks(std::vector<int64_t> *log) { clock_log_ = log; }
    // The whole state as bytes, canonical (members sorted: bucket placement never shows in results, icao_filter.c): equal states
    // <=> equal blobs.  import_state: false = not a state blob (nothing changed).
    void export_state(std::vector<uint8_t> &blob) const;
    bool import_state(const uint8_t *blob, size_t bytes);
    void external_expire() { filter_.expire(); ++nflips_; }   // the host's icaoFilterExpire(), forwarded
    // The serial part: walk the ordered live records of one chunk and decide which frames the
    // reference accepts (best phase, ICAO filter, skip-ahead, filter clock).  Fills acc[0..return) (the
    // vector is only ever grown, to aux_cap entries) and
    // writes each accepted frame's chunk-relative position / skip length / buffer limit (inputs of
    // k_window_stats) to the aux arrays (capacity aux_cap); returns the number of accepted frames,
    // or -1 if aux_cap was too small.  recs[nrecs] must be a readable sentinel with pos = 0xFFFFFFFF.
    int64_t decide(const PhaseRec *recs, uint64_t nrecs, const std::vector<BufferClock> &buffers, std::vector<Accepted> &acc,
                   uint32_t *aux_pos, uint16_t *aux_skip, uint32_t *aux_limit, uint64_t aux_cap, ResolveCounts &counts);

    // The same walk, parallel over buffer ranges and still exact.  The skip window never crosses a
    // buffer, so ranges only interact through the ICAO filter, and the filter only enters through
    // "is this address known right now".  A batch of ranges is walked at once, one thread per range.
    // The first range of a batch starts from the true state, so it simply runs the serial loop on a
    // private copy of the state (adopted afterwards).  The others speculate (spec_walk, const): an
    // address is known iff the filter holds it at the start of the batch, or an earlier range of the
    // batch has a clean DF17 / DF11-IID0 record of it (collect_adders; such a frame is accepted unless
    // something hides it), or the range itself has added it; every address asked about before the
    // range's own first add is remembered, and the range runs the 60 s expiry clock itself when the
    // buffer grid says the expiry cannot have happened before it.  commit_segment (stream order, one
    // thread) checks those assumptions against the true filter at the start of the range and against
    // what expiry / resize dropped, and replays the range's adds and buffer clocks on the true filter
    // when they hold; when they do not, a new batch starts at that range — which is then the exact one.
    // Every range ends up holding the reference's decisions; a batch always completes at least one.
    using Runner = std::function<void(int ntasks, const std::function<void(int)> &task)>;
    void parallel_walk(const PhaseRec *recs, uint64_t nrecs, const std::vector<BufferClock> &buffers, std::vector<SegmentWalk> &segs,
                       const Runner &run, uint64_t *batches = nullptr);
    void union_snapshot(std::vector<uint32_t> &sorted_union) const { filter_.union_sorted(sorted_union); }

    // The stateless part: struct modesMessage fields of the accepted frames (demod_2400.c:399-445,
    // mode_s.c:443-606).  sig[i] = sum of mag^2 over the frame record i would occupy; or msig[n] (when not null) = that of accepted frame n, bit 63 aside.
    static void build_messages(const PhaseRec *recs, const unsigned long long *sig, const unsigned long long *msig, const std::vector<BufferClock> &buffers,
                               const Accepted *acc, uint64_t nacc, mgpu_msg *out);
    // decide + build_messages in one call
    int64_t walk(const PhaseRec *recs, const unsigned long long *sig, uint64_t nrecs,
                 const std::vector<BufferClock> &buffers, std::vector<mgpu_msg> &out, uint32_t *aux_pos,
                 uint16_t *aux_skip, uint32_t *aux_limit, uint64_t aux_cap, ResolveCounts &counts);
    // the per-buffer filter clock for a buffer that produced no walk (zero-length EOF buffer)
    void tick_empty(int64_t sysTimestamp, int64_t sampleTimestamp = kNoBufferTs);
    IcaoFilter &filter() { return filter_; }
    uint64_t nflips() const { return nflips_; }

    // --- the walk on the device (kernels/walk.inc) ---
    int64_t next_flip() const { return next_flip_; }
    // Catch the filter up with what the device walk did — per buffer its first adds in order (summary rows of 6 words:
    // accepted, adds, end clock lo / hi, offset into `adds`, offset into the accept list), then the clock — and tell
    // whether the walk's premises held: the table did not grow (growing empties the other generation, icao_filter.c:65-93),
    // the expiry came after the buffer the walk had put it after (flip; 0x7fffffff = not in this chunk), and only once.
    // On false the state is what it was and the chunk has to be walked here.
    bool apply_device_walk(const uint32_t *per_buf, const uint32_t *adds, uint32_t nbuf, int32_t flip);
    // The device walk's algorithm (kernels/walk.inc) restated on the host — every buffer walked on its own against the state at the
    // start of the chunk plus a table of first adds, iterated until the table reproduces itself, then apply_device_walk — so that
    // its exactness argument is tested without a GPU against the serial walk on streams with arriving aircraft, expiries and table
    // growth (mgpu_selftest_device_walk).  Returns the number of accepted frames, or -1 when the fixed point did not settle in
    // `max_walks` or the premises failed (state untouched: the caller walks the chunk with decide()).
    int64_t device_walk_model(const PhaseRec *recs, uint64_t nrecs, const std::vector<BufferClock> &buffers, std::vector<Accepted> &acc,
                              ResolveCounts &counts, uint32_t max_walks, uint32_t *walks);
    void copy_state(const Resolver &o) { copy_state_from(o); }
    bool same_state(const Resolver &o) const {
        return synthetic_now_ == o.synthetic_now_ && next_flip_ == o.next_flip_ && nflips_ == o.nflips_ && filter_.same_as(o.filter_);
    }

  private:
    void copy_state_from(const Resolver &o);
    void adopt(Resolver &shadow);
    std::unique_ptr<Resolver> shadow_;   // private copy of the state for the first range of a batch
    void collect_adders(const PhaseRec *recs, SegmentWalk &w) const;
    void spec_walk(const PhaseRec *recs, const std::vector<BufferClock> &buffers, SegmentWalk &w) const;
    bool commit_segment(const PhaseRec *recs, const std::vector<BufferClock> &buffers, SegmentWalk &w);
    void serial_segment(const PhaseRec *recs, const std::vector<BufferClock> &buffers, SegmentWalk &w);
    template <class Policy>
    int64_t walk_range(Policy &pol, const PhaseRec *recs, uint64_t rec_lo, const BufferClock *bufs, uint32_t b_lo, uint32_t b_hi,
                       Accepted *out, uint32_t *aux_pos, uint16_t *aux_skip, uint32_t *aux_limit, uint64_t aux_cap,
                       ResolveCounts &c) const;
    static constexpr int64_t kNoBufferTs = INT64_MIN;  // "which buffer this is, is not known": the reference's rule applies
    void after_buffer(int64_t buffer_ts = kNoBufferTs);
    bool scheduled(int64_t buffer_ts) const;           // does the imposed schedule expire the filter after this buffer
    const int64_t *sched_ = nullptr;
    size_t nsched_ = 0;
    uint64_t sched_mismatch_ = 0;
    std::vector<int64_t> *clock_log_ = nullptr;
    IcaoFilter filter_;
    std::vector<uint32_t> chunk_drops_, chunk_news_;   // union changes since begin_chunk
    size_t exp_lo_ = 0, exp_hi_ = 0;                   // ... of them, what the batch's expiry dropped: chunk_drops_[exp_lo_, exp_hi_)
    int64_t synthetic_now_ = 0, next_flip_ = 0;
    uint64_t nflips_ = 0;
};

// ---- one capture walked by several ranks (config 5): what a rank does with its chunks, free of the GPU side (api.cpp:
// mgpu_shard_walk adapts packets to it; mgpu_selftest_shard_walk runs the whole protocol on synthetic record streams) ----
struct ShardWalkPlan {
    uint64_t own_first = 0;                 // first sample of the rank's own range; chunks before it are warm-up
    uint32_t buf_samples = 131072;
    int64_t startup_ms = 0;
    int clock_mode = 0;                     // mgpu_config.filter_clock (0 / 1)
    const int64_t *sched = nullptr;         // imposed expiry schedule: sampleTimestamps of the buffers the filter expires after (must outlive the walk)
    size_t nsched = 0;
    const uint8_t *start_state = nullptr;   // the filter state at own_first (then the warm-up is skipped), or null: start at the first chunk
    size_t start_state_bytes = 0;
};
struct ShardWalkOut {
    std::vector<int64_t> clocks;            // end clock of every buffer of the own range
    std::vector<uint8_t> state_first, state_end;
};
// chunk(i) = (first sample, samples) of the i-th chunk, consecutive; walk(i, own) walks it on `res` (own: the caller also builds
// messages / keeps statistics), returns 0 or an error code that ends the walk.  Returns 0, a walk's code, or -1 with *err set.
int shard_walk_core(Resolver &res, const ShardWalkPlan &plan, size_t nchunks, const std::function<void(size_t, uint64_t &, uint64_t &)> &chunk,
                    const std::function<int(size_t, bool)> &walk, ShardWalkOut &out, const char **err);
// A buffer's end clock (Modes.synthetic_now when backgroundTasks looks at it after the buffer, demod_2400.c:412-414) estimated from
// the records alone: the last candidate with an unconditional record, its best phase by the "known" scores.  Appends one per buffer.
void estimate_end_clocks(const PhaseRec *recs, uint64_t nrecs, const std::vector<BufferClock> &buffers, std::vector<int64_t> &out);
// readsb.c:1227-1231 over a list of end clocks: indices of the buffers the filter expires after
void flip_schedule(const int64_t *end_clock, uint64_t nbuf, int64_t startup_ms, int clock_mode, std::vector<uint64_t> &flip_after);

// The buffers an expiry CAN follow (mask[b] = 1), whatever the data: the k-th expiry follows the first buffer whose end clock reaches T_k,
// T_k + 60 000 <= T_(k+1) < T_k + 60 111 ms (the buffer that reaches T_k starts within 55.6 ms of it, its clock ends within its own 55 ms),
// so T_k lies in a window that widens by 111 ms per expiry and only buffers whose 55 ms touch a window matter to the schedule.
// Buffer b's clock starts at (b * buf_samples * 5) / 12000 + startup_ms (sdr_ifile.c:216).  Returns how many are set.
uint64_t expiry_windows(uint64_t nbuf_total, uint32_t buf_samples, int64_t startup_ms, int clock_mode, uint8_t *mask);

// index of the first of nrecs position-sorted records with position >= pos
uint64_t segment_first_record(const PhaseRec *recs, uint64_t nrecs, uint32_t pos);

}  // namespace mgpu
