// api.cpp — C ABI of libmodes_gpu.so (include/modes_gpu.h): context, device memory, the
// per-feed pipeline  convert -> sweep/slice -> pre-screen -> ordered walk -> signal power,
// and the counters the reference keeps in Modes.stats_current.
//
// The product has no CPU compute path: without a usable HIP device mgpu_create() fails with
// MGPU_E_NODEVICE.
#include <hip/hip_runtime.h>
#include <pthread.h>
#include <dirent.h>
#include <sched.h>

#include <cctype>
#include <algorithm>
#include <atomic>
#include <functional>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <new>
#include <string>
#include <vector>

#include "../../include/modes_gpu.h"
#include "kernels.h"
#include "resolve.h"
#include "tables.h"

using namespace mgpu;

constexpr int kPacketWords = 12;                              // header of a shard packet, 64-bit words: stream position, samples, live records,
constexpr uint64_t kPacketMagic = 0x3354454b4341504dull;      // magic, candidates, phases 4/5, 6/7, 8 tried, conditional-only / unconditional candidates, buffers, 0

namespace {
double wall_ms() {
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}
}  // namespace

// A few persistent helper threads for fork-join over small task counts (the caller takes tasks too).
// Helpers spin briefly before blocking: the forks come every few hundred microseconds while a feed runs.
class Team {
  public:
    ~Team() { stop(); }
    // while *hot is set (a feed is running) idle helpers never block: a core that sleeps between two forks a few hundred
    // microseconds apart drops into a deep C-state, and the wake-up latency then costs more than the work (seen as a
    // 2.4x slower walker stage in the first run on an idle box)
    void start(int helpers, const std::atomic<bool> *hot = nullptr) {
        hot_ = hot;
        for (int i = 0; i < helpers; ++i) threads.emplace_back([this] { loop(); });
    }
    void stop() {
        { std::lock_guard<std::mutex> lk(mu_); quit_ = true; ++gen_; }
        wake_.fetch_add(1, std::memory_order_release);
        cv_work_.notify_all();
        for (auto &t : threads) if (t.joinable()) t.join();
        threads.clear();
    }
    void run(int ntasks, const std::function<void(int)> &fn) {
        if (ntasks <= 0) return;
        if (threads.empty() || ntasks == 1) { for (int i = 0; i < ntasks; ++i) fn(i); return; }
        uint32_t g;
        {
            std::lock_guard<std::mutex> lk(mu_);
            g = ++gen_;
            fn_ = &fn; ntasks_ = ntasks;
            pending_.store(ntasks, std::memory_order_relaxed);
            ticket_.store((uint64_t) g << 32, std::memory_order_release);
        }
        wake_.fetch_add(1, std::memory_order_release);
        cv_work_.notify_all();
        work(g, &fn, ntasks);
        // The helpers' last tasks: while a feed runs the caller POLLS for them too.  Asleep on the condition variable it came back
        // a scheduler wake-up later — tens of microseconds on an idle box, a millisecond on a loaded one, per fork — and the walk
        // forks several times per chunk: a candidate for the "slow mode" in which one stage of one process runs 1.2-8 x slower with
        // nothing else different (profiles/r05_headline_runs.txt, r06_host_4rank.txt), like the stage threads' sleeping GPU waits
        // before it (wait_event_spin).
        if (hot_ && hot_->load(std::memory_order_relaxed)) {
            for (unsigned spin = 0; pending_.load(std::memory_order_acquire) != 0 && spin < (1u << 22); ++spin) {
                __builtin_ia32_pause();
                if ((spin & 255) == 255) sched_yield();
            }
        }
        std::unique_lock<std::mutex> lk(mu_);
        cv_done_.wait(lk, [&] { return pending_.load(std::memory_order_acquire) == 0; });
        fn_ = nullptr;
    }
    std::vector<std::thread> threads;

  private:
    // tasks are handed out through one word that also carries the generation, so a helper that is late
    // leaving generation g can never take a task of generation g+1 with g's function
    void work(uint32_t g, const std::function<void(int)> *fn, int n) {
        int done = 0;
        uint64_t t = ticket_.load(std::memory_order_acquire);
        for (;;) {
            if ((uint32_t) (t >> 32) != g || (int) (uint32_t) t >= n) break;
            if (!ticket_.compare_exchange_weak(t, t + 1, std::memory_order_acq_rel)) continue;
            (*fn)((int) (uint32_t) t);
            ++done;
            t = ticket_.load(std::memory_order_acquire);
        }
        if (done && pending_.fetch_sub(done, std::memory_order_acq_rel) == done) {
            std::lock_guard<std::mutex> lk(mu_);
            cv_done_.notify_all();
        }
    }
    void loop() {
        uint64_t seen = wake_.load(std::memory_order_acquire);
        for (;;) {
            // spin (giving the core away in between: a helper that spins through its time slice starves whatever else the
            // scheduler put on this core); block only when no feed is running
            for (int spin = 0; wake_.load(std::memory_order_acquire) == seen; ++spin) {
                __builtin_ia32_pause();
                if ((spin & 63) == 63) sched_yield();
                if (spin >= 4000 && !(hot_ && hot_->load(std::memory_order_relaxed))) break;
            }
            const std::function<void(int)> *fn;
            int n;
            uint32_t g;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_work_.wait(lk, [&] { return quit_ || wake_.load(std::memory_order_acquire) != seen; });
                if (quit_) return;
                seen = wake_.load(std::memory_order_acquire);
                fn = fn_; n = ntasks_; g = gen_;
            }
            if (fn) work(g, fn, n);
        }
    }
    std::mutex mu_;
    std::condition_variable cv_work_, cv_done_;
    const std::function<void(int)> *fn_ = nullptr;
    int ntasks_ = 0;
    uint32_t gen_ = 0;
    std::atomic<uint64_t> ticket_{0}, wake_{0};
    std::atomic<int> pending_{0};
    const std::atomic<bool> *hot_ = nullptr;
    bool quit_ = false;
};

// One pipeline stage's worth of buffers: a chunk of the stream is converted, swept and pre-screened
// into a slot on the GPU while the worker thread walks the previous chunk's records on the host.
struct Slot {
    // device
    uint16_t *d_mag = nullptr;
    PhaseRec *d_pool = nullptr;
    uint32_t *d_dealer = nullptr;         // k_slice's tile dealer and, behind it, k_sweep's step dealer: 2 x 64 counters, one per 256 bytes (handed back zeroed by k_publish)
    uint32_t *d_pool_used = nullptr, *d_unit_first = nullptr, *d_unit_count = nullptr, *d_unit_live = nullptr;
    uint32_t *d_class_final = nullptr, *d_cand_count = nullptr, *d_sweep_part = nullptr;
    uint16_t *d_cand = nullptr;
    size_t class_bytes = 0;
    // one zero-initialised scratch block per chunk: counters | pool_used | per-buffer sums (1 memset, 1 copy back)
    unsigned long long *d_scratch = nullptr, *h_scratch = nullptr;
    size_t scratch_bytes = 0;
    unsigned long long *d_counters = nullptr, *d_sum_level = nullptr, *d_sum_power = nullptr, *d_win = nullptr, *d_msg_sig = nullptr;
    unsigned long long *d_win_part = nullptr;   // k_window_stats: per-workgroup totals of the chunk's skip windows
    double *d_fsum_level = nullptr, *d_fsum_power = nullptr;
    uint32_t *d_msg_pos = nullptr, *d_msg_limit = nullptr;
    uint16_t *d_msg_len = nullptr, *d_msg_skip = nullptr;
    PhaseRec *d_live = nullptr;          // k_prescreen_write: the surviving records, in stream order ...
    unsigned long long *d_live_sig = nullptr;   // ... and each one's would-be signal power
    unsigned long long *d_live_win = nullptr, *h_live_win = nullptr;   // shard passes (allocated by the first): ... and what its would-be skip window holds (k_window_stats_t<true>)
    // pinned host
    PhaseRec *h_live = nullptr;          // their copies: the fetcher pulls exactly nlive records over the copy engine (a kernel storing
    unsigned long long *h_live_sig = nullptr;   // into page-locked host memory waited 64 us per chunk on PCIe write latency)
    hipEvent_t ev_window = nullptr;       // k_window_stats of this slot's last use has run (stream2)
    bool window_pending = false;
    unsigned long long *h_counters = nullptr, *h_sums = nullptr, *h_win = nullptr, *h_sig = nullptr;
    double *h_fsums = nullptr;
    double *d_fsx = nullptr, *h_fsx = nullptr;   // SC16 formats: the float sums' own device / page-locked buffers (k_fsum_sc16 runs beside the chunk and ends on its own)
    uint32_t *h_msg_pos = nullptr, *h_msg_limit = nullptr;
    uint16_t *h_msg_len = nullptr, *h_msg_skip = nullptr;
    hipEvent_t ev_done = nullptr;        // the chunk is complete: recorded behind every chunk, WITHOUT a timestamp (a timed event is a marker the next kernel waits for)
    hipEvent_t ev[5] = {};               // 3: the end of the post-sweep stage (timed chunks only) | stage timing, sampled chunks only (timed): 0 1 convert, 1 4 k_sweep, 4 2 k_slice, 2 3 post-sweep (a timing event costs ~4.5 us of idle stream: neighbouring brackets share theirs)
    bool timed = false;
    uint32_t slice_blocks = 0;            // rows of d_sweep_part the chunk's k_slice wrote
    uint32_t sweep_blocks = 0;            // grid of the chunk's k_sweep
    uint32_t *d_ac_noise = nullptr;       // Mode A/C: per-buffer noise level
    AcCand *h_ac = nullptr;               // ... candidates, written by k_modeac straight into pinned host memory
    hipEvent_t ev_h2d = nullptr;          // the chunk's IQ samples have arrived in HBM (copy stream)
    // SC16 formats: the per-buffer float sums run beside the chunk's kernels on stream2 (k_fsum_sc16): what the converter waited
    // for | the sums are there | the converter has read the samples too
    hipEvent_t ev_pre = nullptr, ev_fsum = nullptr, ev_convdone = nullptr;
    bool fsum_pending = false;
    const uint8_t *fsum_iq = nullptr;     // the chunk's IQ samples (SC16 formats), for the float sums enqueued behind k_sweep
    int fsum_idx = -1;                    // which entry of mgpu_ctx::fsum_ring holds the chunk's float sums
    hipEvent_t ev_scan = nullptr;         // pre-screen offsets are final (main stream) -> the write pass may start (second stream)
    // the second stream keeps out of k_sweep's way (walk_job: hold_behind_sweep): the chunk's k_sweep has run | which chunk that was
    hipEvent_t ev_swept = nullptr;
    std::atomic<uint64_t> swept_seq{~0ull};
    // the converter beside the previous chunk's k_slice (mgpu_ctx::conv_side): its own stream -> the chunk's magnitudes are there
    // (the main stream waits for it) | start of the chunk's k_sweep on the main stream (stage timing: ev[1] is on the converter's stream then)
    hipEvent_t ev_conv = nullptr, ev_sweep0 = nullptr;
    // converter and sweep in one kernel (mgpu_ctx::sweep_fused, k_sweep_uc8): the chunk's samples and the 326 magnitudes before it, as
    // enqueue_convert found them (no converter launch); the per-step sums the kernel leaves for k_slice's prologue
    const uint8_t *fused_iq = nullptr;
    const uint16_t *fused_tail = nullptr;
    uint64_t seq = 0;                     // the chunk's number in the context's life (slot = seq % kSlots)
    // the job
    uint64_t n = 0, stream_pos = 0;
    uint8_t *d_wk_in = nullptr;           // the walk on the device: its input blob (read again by k_build_messages: the buffer clocks) ...
    void *d_wk_acc = nullptr;             // ... the chunk's ordered accept list ...
    unsigned long long *d_wk_sig = nullptr;   // ... and per accepted frame the signal power | long flag
    uint8_t *h_blob = nullptr, *d_blob = nullptr;   // device-messages mode: the walker's accept list + buffer clocks, page-locked host / device
    bool sig_late = false;                // the signal powers of this chunk are computed after the walk, for the accepted frames (k_msg_sig)
    int feed = -1;                        // deferred feeds: which FeedSlot the chunk's messages go to (-1: mgpu_ctx::pending)
    int32_t thr = 58;                     // preamble threshold of this chunk (raised after drops, demod_2400.c:335-338)
    bool have_mag = false, busy = false;
    bool have_noise = false;              // mag_buf entry with the caller's mean_level: Mode A/C noise level computed on the host
    uint32_t given_noise = 0;
    std::vector<BufferClock> buffers;
    std::vector<double> given_mean_power;
};

// Decoded messages waiting for mgpu_collect: a 64-byte aligned array that grows geometrically and is
// never value-initialised (the builder writes every byte of every message with streaming stores).
struct MsgBuf {
    mgpu_msg *p = nullptr;
    size_t n = 0, cap = 0;
    bool external = false;                            // p is the caller's buffer (mgpu_set_message_buffer): never grown, never freed
    ~MsgBuf() { if (!external) free(p); }
    size_t size() const { return n; }
    mgpu_msg *data() { return p; }
    void clear() { n = 0; }
    bool grow_for(size_t extra) {                     // room for `extra` more messages
        if (cap - n >= extra) return true;
        if (external) return false;
        size_t want = n + extra;
        if (want < 2 * cap) want = 2 * cap;
        void *q = nullptr;
        if (posix_memalign(&q, 64, want * sizeof(mgpu_msg)) != 0) return false;
        if (n) std::memcpy(q, p, n * sizeof(mgpu_msg));
        free(p);
        p = (mgpu_msg *) q;
        cap = want;
        return true;
    }
    void drop_front(size_t k) {
        if (k < n) std::memmove(p, p + k, (n - k) * sizeof(mgpu_msg));
        n -= k;
    }
    void use_external(mgpu_msg *buf, size_t capacity) {
        if (!external) free(p);
        p = buf; cap = capacity; n = 0; external = buf != nullptr;
        if (!external) { p = nullptr; cap = 0; }
    }
};

// What the builder thread needs of a chunk once its slot has gone back to the GPU.
struct HostJob {
    std::vector<PhaseRec> recs;              // the chunk's live records (heap copy of Slot::h_live)
    std::vector<unsigned long long> sig;
    std::vector<unsigned long long> win;     // shard passes: per live record the packed counts of its would-be skip window
    std::vector<Accepted> acc;               // the walker's decisions
    std::vector<uint32_t> pos;               // their chunk-relative scan positions
    std::vector<BufferClock> buffers;
    std::vector<double> given_mean_power;
    std::vector<unsigned long long> sums;    // per-buffer level / power sums of the converter
    std::vector<double> fsums;
    int fsum_idx = -1;                       // >= 0: the float sums are still on their way (mgpu_ctx::fsum_ring): the builder waits for them, not the fetcher
    std::vector<AcCand> ac;                  // Mode A/C candidates of the chunk (cfg.mode_ac)
    ResolveCounts rc;
    uint64_t nlive = 0;
    uint32_t nmsg = 0;                       // accepted frames: acc[0..nmsg), pos[0..nmsg)
    uint64_t stream_pos = 0;                 // stream position of the chunk's first sample
    int slot = -1;                           // the slot the chunk ran in (the walker still needs its device side)
    int feed = -1;                           // Slot::feed
    bool busy = false;
    // the walk ran on the device (MGPU_DEVICE_WALK=1): no records here; per message the signal power (bit 63: a 112-bit frame as
    // sliced) and — unless the messages stay on the device — the records k_build_messages made, both copied into page-locked memory
    bool from_device = false;
    bool sig_late = false;                   // Slot::sig_late: sig[] is empty, h_msig holds the accepted frames' signal powers (ev_copied)
    bool fetched = false;                    // recs / sig hold the chunk's live records
    mgpu_msg *h_msgs = nullptr;
    unsigned long long *h_msig = nullptr;
    hipEvent_t ev_copied = nullptr;          // ... the copies have landed (stream2)
    std::vector<uint32_t> buf_nacc;          // accepted frames per buffer
};

// Deferred feeds (mgpu_set_deferred): a feed call returns once its chunks are enqueued, the next one may follow at once, and
// mgpu_collect waits for the oldest uncollected feed only.  Each feed in flight has its own message list.
struct FeedSlot {
    MsgBuf msgs;
    // device-messages mode (mgpu_set_device_messages): the feed's messages are built by k_build_messages into d_msgs
    mgpu_msg *d_msgs = nullptr;
    mgpu_msg *d_ext = nullptr;                // mode 1: the caller's own device buffer for this feed's records (mgpu_set_device_message_buffer), else d_msgs
    uint64_t d_ext_cap = 0;
    mgpu_msg *d_list = nullptr;               // ... whichever of the two this feed's k_build_messages write to,
    uint64_t d_list_cap = 0;                  // ... and the records it holds
    mgpu_msg *host_dev = nullptr;             // mode 2 (device-built, host-delivered): the device address of msgs.p, the caller's page-locked array
    uint64_t d_cap = 0, d_count = 0;          // d_count: walker thread only, read by the caller after the feed is complete
    hipEvent_t ev_built = nullptr;            // the last k_build_messages of the feed has run (stream2)
    uint64_t jobs_total = 0, jobs_built = 0;  // chunks submitted / chunks whose messages are complete (under mgpu_ctx::mu)
    bool closed = false;                      // every chunk of the feed has been submitted
};

struct mgpu_ctx {
    // Ten slots (round 6; four in rounds 4-5, three before): a slot is held from the moment the feeding thread enqueues the chunk's
    // kernels until its walk is done.  With four, a caller that keeps two feeds of four chunks in flight (feed k + 1 before
    // collect k: bench.py, the C hosts) spent most of every feed call waiting for a slot, the GPU's queue was never more than one
    // or two chunks deep, and every hiccup of a host stage was a bubble on the GPU: 1.42-1.44 ms per 537 M samples with 4, 5 or 6
    // slots, 1.335 — the kernels' sum — with 8, 10 or 12 (profiles/r06_slots.txt).  Ten = the eight chunks of two feeds + two of
    // slack; ~1.5 GB of HBM each at the default chunk size, out of 288.
#ifndef MGPU_SLOTS
#define MGPU_SLOTS 10
#endif
    static constexpr int kSlots = MGPU_SLOTS;
    static constexpr int kJobs = MGPU_SLOTS + 2;              // fetched -> walked -> built: a job outlives its slot by the builder's stage
    static constexpr int kFsumRing = 2 * MGPU_SLOTS + 4;      // > kSlots + kJobs: an entry is free again before its index comes round
    mgpu_config cfg{};
    hipStream_t stream = nullptr, stream2 = nullptr, stream_w = nullptr;   // main | window statistics | pre-screen write pass / IQ uploads
    hipStream_t stream_d2h = nullptr;                                      // the fetcher's record copies
    // SC16 formats: the float sums of the chunks in flight — a ring, not the slots' own buffers, so that nobody has to wait for a chunk's
    // sums before the chunk's slot goes back to the GPU (chunk seq uses entry seq % kFsumRing)
    struct FsumRing { double *d = nullptr, *h = nullptr; void *scratch = nullptr; hipEvent_t ev = nullptr; } fsum_ring[kFsumRing];
    uint32_t prescreen_variant = 3;                                        // PostSweepParams::variant (the experiments build can ask for the older passes)
    uint32_t cu_mask[32] = {0}, cu_mask_words = 0;                         // the side streams' CU mask (every CU: mgpu_create says what it is for)
    int post_beside = 0;                                                   // experiment: 1 = stream_pw takes the write pass, 2 = the count pass too
    hipStream_t stream_pw = nullptr;                                       // experiment (MGPU_WRITE_BESIDE=1, experiments build): the pre-screen's write pass + k_publish on a stream of their own, beside the next chunk's converter — measured 297 against 357 Gsamples/s (gpurun r05i): beside a kernel that saturates the memory system the write pass's dependent round trips stretch the post-sweep stage from 0.25 to 0.73 ms per step.  Round 6, beside k_sweep_uc8 (1 / 2: + the count pass; 3 / 4: on a mask-API stream): 1.27-1.39 ms per feed against 1.21-1.25; the whole stage held back until the next chunk's sweep is through, beside its k_slice: 1.30-1.36 (profiles/r06_sweep_fused.txt)
    hipStream_t stream_f = nullptr;                                        // SC16 formats: the float sums' chains (k_fsum_sc16), so that what follows a walk does not queue behind them
    // The UC8 converter of chunk N + 1 beside chunk N's k_slice (round 6): the converter is the pipeline's one HBM-bound
    // kernel, k_slice its largest issue-bound one.  stream_c carries the converters, each held behind the k_sweep of the chunk before;
    // k_slice's grid is capped at three workgroups per CU (slice_blocks_cap) so that a converter workgroup (32 KB of LDS) fits beside them.
    hipStream_t stream_c = nullptr;
    int d2h_hold = 0;                                                      // the fetcher's record copy held behind the next chunk's pending k_sweep (fetch_records); 0: experiments build, MGPU_D2H_HOLD
    int s2_hold = 1;                                                       // the second stream's work held behind a pending k_sweep (hold_behind_sweep); 0: experiments build, MGPU_S2_HOLD
    int sweep_fused = 3;                                                   // without Mode A/C the sweep converts on the way (no converter launch, the magnitudes written once) — bit 0: UC8, k_sweep_uc8; bit 1: SC16 / SC16Q11, k_sweep_sc16; 0: k_convert_* + k_sweep (experiments build: MGPU_SWEEP_FUSED)
    int conv_side = 0;                                                     // 1: on (UC8 without Mode A/C, 1-bit repair tables: with the 2-bit tables k_slice's three workgroups leave no LDS)
    int convert_variant = 0;                                               // launch_convert's variant (1: the round-1..5 converter; experiments build)
    unsigned conv_side_blocks = 2048, slice_blocks_cap = 0;                // grid of the side converter | of k_slice beside it (0: whatever is resident)
    hipStream_t s_post = nullptr;                                          // what follows the walk (window statistics, messages on the device): stream2, or stream_wk
    hipStream_t stream_wk = nullptr;                                       // the walk on the device: highest priority, its small kernels must not queue behind the main stream's
    std::string err;

    uint64_t cap_samples = 0;      // per feed call (cfg.max_samples)
    uint64_t chunk_samples = 0;    // per pipeline slot
    uint64_t cap_units = 0, cap_buffers = 0, cap_pool = 0, cap_msgs = 0, cap_ac = 0;   // per slot

    uint8_t *d_iq = nullptr;
    // host feeds upload chunk i of a feed into region i of d_iq (copy stream); the converter that read region i last (main stream)
    // must have run before the next upload into it may start — with deferred feeds of one or two chunks nothing else orders them
    std::vector<hipEvent_t> ev_iq_read;   // per region: recorded behind the converter of the last chunk uploaded there
    std::vector<char> iq_region_used;
    const uint16_t *tail_src = nullptr;   // device: the 326 magnitudes before the next chunk (end of the previous chunk's d_mag)
    uint64_t chunk_seq = 0;               // chunks alternate between the two slots across feeds
    uint32_t *d_adder_bitmap = nullptr;
    uint32_t *d_bit_syndrome = nullptr, *d_group_syndrome = nullptr;
    uint64_t *d_parity = nullptr, *d_tab_long = nullptr, *d_tab_short = nullptr;
    uint16_t *d_uc8_folded = nullptr;
    int n_long = 0, n_short = 0;
    Slot slot[kSlots];
    unsigned long long *d_win = nullptr, *h_win = nullptr;   // skip-window totals of the current feed
    uint64_t feed_cand[8] = {0, 0, 0, 0, 0, 0, 0, 0};         // C, phase[5], U, R of the current feed
    ResolveCounts feed_rc;
    std::vector<uint32_t> w_limit;                            // walker scratch (ordinary memory)
    std::vector<uint16_t> w_skip;
    HostJob job[kJobs];                                         // fetcher -> walker -> builder hand-off ring
    uint64_t job_seq = 0;

    std::vector<SyndromeEntry> tab_long, tab_short;
    uint32_t valid_long = 0, valid_short = 0;

    Resolver resolver;
    MsgBuf pending;
    static constexpr int kFeeds = 4;
    FeedSlot feed[kFeeds];                                    // deferred mode: ring of feeds in flight / uncollected
    uint64_t feed_head = 0, feed_tail = 0;                    // oldest uncollected feed, next feed to open
    bool deferred = false;
    int device_msgs = 0;                                      // mgpu_set_device_messages: 1 = the records stay in HBM (FeedSlot::d_msgs), 2 = k_build_messages stores them into the caller's page-locked array
    bool sig_late = true;                                     // MGPU_SIG_LATE=0: signal power of every live record in the pre-screen write pass (as in shard passes) instead of the accepted frames' after the walk
    int timing_every = 15;                                    // chunks per set of stage timing events (1 = every chunk; MGPU_TIMING_EVERY in the experiments build).  Odd: with feeds of four chunks the sampled chunk is not always a feed's first
    bool fsum_wide = false;                                   // (experiments build: MGPU_FSUM_WIDE=1) the float sums as three wide kernels instead of one chain per buffer
    float event_bracket_us = 4.5f;                            // what a pair of timing events adds to the kernel it brackets (mgpu_event_bracket_us measures it)
    uint64_t timing_seq = 0;
    bool accounting_open = false;                             // feed_begin has run, feed_end has not (deferred: spans several feeds)
    double acct_t0 = 0;
    mgpu_counters counters{};
    mgpu_timing timing{}, acc{};
    uint64_t stream_pos = 0;
    bool eof = false;

    // host pipeline behind the GPU: the walker thread takes the slots in submission order (record copy,
    // ordered accept walk, window-statistics launch) and hands a HostJob to the builder thread
    // (messages, signal / noise statistics), so that the serial walk is all the walker does
    std::thread fetcher, worker, builder;
    Team walk_team, build_team;                               // helpers of the walker / builder stage (MGPU_WALK_THREADS, MGPU_BUILD_THREADS)
    int walk_threads = 4, build_threads = 3;
    int walk_ranges = 0;                                      // buffer ranges per round of the host's walk (0: one per walk thread)
    std::atomic<bool> hot{false};                             // a feed is running: the stage threads and helpers poll instead of sleeping
    std::vector<int> host_cpus;                               // the CPUs the host threads were pinned to (empty: not pinned)
    std::vector<SegmentWalk> segs;                            // the walker's buffer ranges
    std::vector<mgpu_msg> b_stage;                            // builder scratch (Mode A/C merge)
    // time-sharded capture (config 5, mgpu_shard_*): 0 = normal, 1 = sweep for the adder bitmap only, 2 = keep the
    // pre-screened records of every chunk as packets instead of walking them
    int shard_mode = 0;
    std::vector<uint8_t> shard_packets;
    // the sharded walk (mgpu_shard_walk): the imposed expiry schedule (the resolver points into it), the range's end clocks, what
    // each of its buffers adds to noise_power_sum, the filter state at the range's first sample / at its end
    std::vector<int64_t> shard_sched;
    std::vector<int64_t> shard_est;                           // the fetcher's estimate of every buffer's end clock, packet by packet
    std::vector<uint64_t> shard_est_pos, shard_est_off;       // ... the packets' first samples / offsets into shard_est
    std::vector<double> shard_noise;
    std::vector<uint64_t> shard_sig;                          // ... and every accepted message's sum of squared magnitudes (its signal power's numerator): 8 bytes
                                                              // per message for the sum blocks, where the messages themselves are 64
    ShardWalkOut shard_out;
    bool shard_noise_on = false;                              // a rank's pass through the ordinary pipeline (mgpu_shard_stream_*): the builder logs every buffer's noise term
    uint64_t shard_stream_own_first = 0;
    bool shard_stream = false, shard_stream_cold = false;
    bool shard_marked = false;                                // ... the range has begun for the walker (shard_mark_now): with deferred feeds the walker gets there on its own
    // beast encoder scratch (mgpu_beast_encode*): grown on demand
    uint16_t *d_beast_len = nullptr;        // per message: frame length | signal byte << 8
    uint8_t *d_beast_in = nullptr, *d_beast_out = nullptr;
    unsigned long long *d_beast_off = nullptr;
    int device_slot = -1;                                     // which of the device's pipeline core groups this context pinned to
    // the ordered walk on the device (kernels/walk.inc).  MGPU_DEVICE_WALK=1: the walker thread only checks the walk's premises
    // and catches the filter up (Resolver::apply_device_walk), the chunk's records stay in HBM; =check: beside the host walk,
    // every decision compared (mgpu_debug_device_walk)
    int device_walk = 0;                                      // 0 off, 1 on, 2 check
    bool wk_serial_only = false;                              // MGPU_DBG_WK_SERIAL: every buffer through k_walk's serial decision loop (cross-check of the lane-parallel one)
    WalkBuffers wk{};
    uint8_t *h_wk_in = nullptr, *h_wk_sum = nullptr;
    size_t wk_in_cap = 0;
    mgpu_msg *d_wk_msgs = nullptr;                            // k_build_messages' output when the messages go to the host
    uint32_t wk_acc_cap = 0;                                  // accepted frames per buffer the walk has room for
    hipEvent_t ev_wk = nullptr;
    Resolver wk_shadow;                                       // check mode: the state before the host walk, for apply_device_walk
    uint64_t wk_stats[8] = {0, 0, 0, 0, 0, 0, 0, 0};          // chunks, taken from the device, not converged, premises failed (host walk), refused, walks, mismatches, -
    mgpu_fields *d_fields = nullptr;
    uint64_t fields_cap = 0;
    double *d_roll_tan = nullptr;                             // tables.h build_roll_tangent_table(), uploaded on first use
    // the first-stage tracking gate (kernels/gate.inc): the aircraft table (1 GiB, allocated and zeroed by the first call), its scratch
    void *d_gate_table = nullptr, *d_gate_scratch = nullptr;
    uint8_t *d_gate_verdict = nullptr;
    uint64_t gate_cap = 0;
    uint32_t *d_beast_blocks = nullptr;
    mgpu_deferred *d_deferred = nullptr;                      // mgpu_beast_encode_gated's list, device side
    uint64_t deferred_cap = 0;
    hipStream_t stream_aux = nullptr;                         // field decode / beast encoder / tracking gate: synchronous calls, not behind the pipeline's queued chunks
    unsigned long long *d_beast_total = nullptr;
    uint64_t beast_cap_msgs = 0, beast_cap_in = 0, beast_cap_out = 0;
    uint16_t *d_hist = nullptr;                               // magnitudes of the 326 samples before the shard
    uint8_t *d_hist_iq = nullptr;
    unsigned long long *d_hist_sums = nullptr;
    // experiment / debug switches, read once at creation
    bool dbg_print = false;
    int dbg_stage = 0;
    std::string dump_dir;
    double feed_t0 = 0;                                       // wall clock at feed start (MGPU_DEBUG_PRINT timeline)
    uint64_t spec_segments = 0, spec_batches = 0;            // ranges walked, batches it took
    std::mutex mu;
    std::condition_variable cv;
    std::deque<int> queue, walk_queue, build_queue;
    bool stop = false;
    int worker_rc = MGPU_OK;
};

#define HIPCHK(ctx, call)                                                                          \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                        \
            return e_ == hipErrorOutOfMemory ? MGPU_E_NOMEM : MGPU_E_HIP;                          \
        }                                                                                          \
    } while (0)

// Wait for `pred` (evaluated under c->mu).  During a feed: poll with the lock released and the core offered to others;
// otherwise block on the condition variable.
template <class Pred>
static void stage_wait(mgpu_ctx *c, std::unique_lock<std::mutex> &lk, Pred pred) {
    while (!pred()) {
        if (c->hot.load(std::memory_order_relaxed)) {
            lk.unlock();
            for (int i = 0; i < 32; ++i) __builtin_ia32_pause();
            sched_yield();
            lk.lock();
        } else {
            c->cv.wait(lk);                      // (feed_begin notifies after raising `hot`)
        }
    }
}

// Waiting for the GPU on a stage thread.  hipEventSynchronize / hipStreamSynchronize spin for a short while and then go to sleep on an
// interrupt; on the pool's (shared, loaded) hosts the wake-up came back ~1.0 ms later, every time — and the pipeline has two steady
// states: when the builder reaches its wait for a chunk's signal powers a little early it sleeps, is 1 ms late for the next chunk
// too, and the step takes 3-14 ms instead of 1.5 (3 of 14 back-to-back benchmark runs, gpurun r05o: "build_host" 14 ms per step,
// every stage and kernel as fast as ever — round 4's unexplained "a single host stage now and then runs 2-4 x slower").  The stage
// threads poll for their jobs anyway, so they poll for the GPU as well: hipEventQuery / hipStreamQuery read the
// completion signal in memory, no system call, ~2 us between looks.
static hipError_t wait_event_spin(hipEvent_t ev) {
    for (unsigned spin = 0;; ++spin) {
        const hipError_t e = hipEventQuery(ev);
        if (e != hipErrorNotReady) return e;
        for (int k = 0; k < 128; ++k) __builtin_ia32_pause();
        if ((spin & 1023) == 1023) sched_yield();
    }
}
static hipError_t wait_stream_spin(hipStream_t s) {
    for (unsigned spin = 0;; ++spin) {
        const hipError_t e = hipStreamQuery(s);
        if (e != hipErrorNotReady) return e;
        for (int k = 0; k < 128; ++k) __builtin_ia32_pause();
        if ((spin & 1023) == 1023) sched_yield();
    }
}

static int fetch_slot(mgpu_ctx *c, Slot &sl, HostJob &job, Slot *next);
static int walk_job(mgpu_ctx *c, Slot &sl, HostJob &job);
static int build_job(mgpu_ctx *c, HostJob &job);
static void fetcher_main(mgpu_ctx *c);
static void worker_main(mgpu_ctx *c);
static void builder_main(mgpu_ctx *c);

// Put the host threads next to the device and next to each other: on the GPU's NUMA node (the
// record buffers are pinned host memory the GPU writes over PCIe, allocated there), and on
// physical cores that share one L3 — each stage reads what the previous one has just written, and a cross-CCD hand-off costs a fabric round trip per cache line.  The L3 group is
// picked by device ordinal so that the ranks of a node do not pile onto one CCD.
// MGPU_NO_AFFINITY=1 leaves the threads unbound.
static int sysfs_int(const std::string &path, int dflt) {
    FILE *f = fopen(path.c_str(), "r");
    if (!f) return dflt;
    int v = dflt;
    if (fscanf(f, "%d", &v) != 1) v = dflt;
    fclose(f);
    return v;
}

// Several contexts of one process on the same device (fan-in: one context per sample stream) must not pin their pipelines
// onto the same cores: the k-th live context of a device takes another L3 group (below).
static std::mutex g_slot_mu;
static uint32_t g_device_slots[64];          // bit k set = the device's k-th pipeline slot is taken

static int take_device_slot(int device) {
    std::lock_guard<std::mutex> lk(g_slot_mu);
    uint32_t &m = g_device_slots[device & 63];
    for (int k = 0; k < 32; ++k)
        if (!(m & (1u << k))) { m |= 1u << k; return k; }
    return 0;
}

static void release_device_slot(int device, int slot) {
    std::lock_guard<std::mutex> lk(g_slot_mu);
    g_device_slots[device & 63] &= ~(1u << slot);
}

// What the PROCESS may run on (cgroups, taskset): the mask of the thread that loaded the library, taken once, at load time.  Not the
// calling thread's mask of the moment: an application that follows mgpu_host_cpus' advice keeps its own threads — the one that
// creates the next context included — OFF the first context's cores, and a second context that picked its cores from that
// thread's mask landed on other L3 groups, its walk 2-3 x slower (bench.py's extra configurations, rounds 2 and 3).
static cpu_set_t g_process_cpus;
static bool g_process_cpus_ok = false;
__attribute__((constructor)) static void remember_process_cpus() { g_process_cpus_ok = sched_getaffinity(0, sizeof(g_process_cpus), &g_process_cpus) == 0; }
static bool process_cpus(cpu_set_t *out) {
    if (g_process_cpus_ok) { *out = g_process_cpus; return true; }
    return sched_getaffinity(0, sizeof(*out), out) == 0;
}

// CPUs of the device's NUMA node that the process may use (empty set: unknown)
static bool device_local_cpus(int device, cpu_set_t *out) {
    CPU_ZERO(out);
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, (int) sizeof(bus), device) != hipSuccess) return false;
    std::string id(bus);
    for (auto &ch : id) ch = (char) tolower((unsigned char) ch);
    FILE *f = fopen(("/sys/bus/pci/devices/" + id + "/local_cpulist").c_str(), "r");
    if (!f) return false;
    char line[4096] = {0};
    const bool ok = fgets(line, sizeof(line), f) != nullptr;
    fclose(f);
    if (!ok) return false;
    cpu_set_t allowed;
    if (!process_cpus(&allowed)) return false;
    int n = 0;
    for (char *tok = strtok(line, ",\n"); tok; tok = strtok(nullptr, ",\n")) {
        int a = 0, b = 0;
        const int got = sscanf(tok, "%d-%d", &a, &b);
        if (got == 1) b = a;
        if (got >= 1)
            for (int k = a; k <= b && k < CPU_SETSIZE; ++k)
                if (CPU_ISSET(k, &allowed)) { CPU_SET(k, out); ++n; }
    }
    return n > 0;
}

// Page-locked host memory is placed where the allocating thread runs: while the context allocates its buffers (the record
// copies' destinations, the counter blocks) the calling thread sits on the device's NUMA node, whatever CPU it came from —
// on a two-socket box a process that happened to start on the other socket had every pipeline stage read its records across
// the socket link (2.3 vs 2.6 ms per step from run to run).
struct NearDevice {
    cpu_set_t saved;
    bool moved = false;
    explicit NearDevice(int device) {
        if (getenv("MGPU_NO_AFFINITY")) return;
        cpu_set_t local;
        if (sched_getaffinity(0, sizeof(saved), &saved) != 0 || !device_local_cpus(device, &local)) return;
        moved = pthread_setaffinity_np(pthread_self(), sizeof(local), &local) == 0;
    }
    ~NearDevice() { if (moved) (void) pthread_setaffinity_np(pthread_self(), sizeof(saved), &saved); }
};

static std::string sysfs_line(const std::string &path) {
    char line[4096] = {0};
    FILE *f = fopen(path.c_str(), "r");
    if (!f) return std::string();
    const bool ok = fgets(line, sizeof(line), f) != nullptr;
    fclose(f);
    std::string v = ok ? line : "";
    while (!v.empty() && (v.back() == '\n' || v.back() == ' ')) v.pop_back();
    return v;
}

// index of the PCI function `id` ("0000:c1:00.0") among the functions with its vendor, device id and local CPU list, ordered by address; -1: unknown
static int device_index_on_node(const std::string &id, const std::string &base = "/sys/bus/pci/devices/") {
    const std::string vendor = sysfs_line(base + id + "/vendor"), dev = sysfs_line(base + id + "/device"), cpus = sysfs_line(base + id + "/local_cpulist");
    if (vendor.empty() || dev.empty() || cpus.empty()) return -1;
    DIR *d = opendir(base.c_str());
    if (!d) return -1;
    std::vector<std::string> same;
    while (const dirent *e = readdir(d)) {
        const std::string name = e->d_name;
        if (name.empty() || name[0] == '.') continue;
        if (sysfs_line(base + name + "/vendor") == vendor && sysfs_line(base + name + "/device") == dev && sysfs_line(base + name + "/local_cpulist") == cpus)
            same.push_back(name);
    }
    closedir(d);
    std::sort(same.begin(), same.end());
    const auto it = std::find(same.begin(), same.end(), id);
    return it == same.end() ? -1 : (int) (it - same.begin());
}

// Two groups of threads, two L3 groups: `walk` (the walker and its helpers: they pass cache lines of the filter state and of
// the record list among themselves all the time) and `rest` (fetcher, builder and its helpers).  The hand-off between the
// two — one job per chunk — crosses CCDs once.  An 8-GPU node has two CCDs per GPU on the GPU's own NUMA node (EPYC 9575F:
// 16 L3 groups, 8 per socket, 4 GPUs per socket), so the first context of every device gets two groups of its own; further
// contexts of the same device (fan-in) put both groups of threads on one L3 group, half the node's groups away.
static void bind_near_device(std::thread *const *walk, int nwalk, std::thread *const *rest, int nrest, int device, int device_slot,
                             std::vector<int> *pinned) {
    if (getenv("MGPU_NO_AFFINITY")) return;
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, (int) sizeof(bus), device) != hipSuccess) return;
    std::string id(bus);
    for (auto &ch : id) ch = (char) tolower((unsigned char) ch);
    FILE *f = fopen(("/sys/bus/pci/devices/" + id + "/local_cpulist").c_str(), "r");
    if (!f) return;
    char line[4096] = {0};
    const bool ok = fgets(line, sizeof(line), f) != nullptr;
    fclose(f);
    if (!ok) return;
    cpu_set_t allowed;                     // never step outside what the process may use (cgroups, taskset)
    if (!process_cpus(&allowed)) return;
    std::vector<int> cpus;
    for (char *tok = strtok(line, ",\n"); tok; tok = strtok(nullptr, ",\n")) {
        int a = 0, b = 0;
        const int got = sscanf(tok, "%d-%d", &a, &b);
        if (got == 1) b = a;
        if (got >= 1)
            for (int k = a; k <= b && k < CPU_SETSIZE; ++k)
                if (CPU_ISSET(k, &allowed)) cpus.push_back(k);
    }
    if (cpus.empty()) return;
    // L3 groups of the node, in first-appearance order
    std::vector<int> l3_ids, l3_of(cpus.size());
    for (size_t i = 0; i < cpus.size(); ++i) {
        l3_of[i] = sysfs_int("/sys/devices/system/cpu/cpu" + std::to_string(cpus[i]) + "/cache/index3/id", -1);
        if (std::find(l3_ids.begin(), l3_ids.end(), l3_of[i]) == l3_ids.end()) l3_ids.push_back(l3_of[i]);
    }
    // Which of the node's GPUs this is: its place among the PCI functions of the same vendor / device id on the same NUMA node, by bus
    // address (sysfs is not namespaced: a container that was handed ONE of a node's eight GPUs still sees the others there).  The HIP
    // ordinal is 0 in every such container — four of them on one socket picked the same two L3 groups and the same cores (round 5:
    // now and then a benchmark process ran at 0.6 of the rate with one host stage slow and nothing else changed).  Ranks that
    // share one device (tests) or see one device each (per-rank HIP_VISIBLE_DEVICES) still spread out by LOCAL_RANK.
    int ordinal = device_index_on_node(id);
    if (ordinal < 0) ordinal = device;
    if (const char *lr = getenv("LOCAL_RANK")) { const int v = atoi(lr); if (v >= 0) ordinal = v; }
    const size_t ng = l3_ids.size();
    int want_walk, want_rest;
    if (device_slot == 0 && ng >= 2) {
        want_walk = l3_ids[((size_t) ordinal * 2) % ng];
        want_rest = l3_ids[((size_t) ordinal * 2 + 1) % ng];
    } else {   // further contexts of the same device: half the node's groups away, where an 8-GPU node's other devices do not sit
        const size_t stride = ng >= 2 ? ng / 2 : 1;
        want_walk = want_rest = l3_ids[((size_t) ordinal * 2 + (size_t) device_slot * stride + (size_t) (device_slot / 2)) % ng];
    }
    // one logical CPU per physical core of a group; more threads than cores share cores round-robin
    auto cores_of = [&](int want) {
        std::vector<int> pick, cores;
        for (size_t i = 0; i < cpus.size(); ++i) {
            if (l3_of[i] != want) continue;
            const int core = sysfs_int("/sys/devices/system/cpu/cpu" + std::to_string(cpus[i]) + "/topology/core_id", (int) i);
            if (std::find(cores.begin(), cores.end(), core) != cores.end()) continue;
            cores.push_back(core);
            pick.push_back(cpus[i]);
        }
        return pick;
    };
    const std::vector<int> pw = cores_of(want_walk), pr = cores_of(want_rest);
    cpu_set_t set;
    if (want_walk >= 0 && pw.size() >= 2 && pr.size() >= 2) {
        const bool same = want_walk == want_rest;
        for (int t = 0; t < nwalk; ++t) {
            CPU_ZERO(&set); CPU_SET(pw[(size_t) t % pw.size()], &set);
            (void) pthread_setaffinity_np(walk[t]->native_handle(), sizeof(set), &set);
        }
        for (int t = 0; t < nrest; ++t) {       // on a shared group the second set of threads continues where the first ended
            CPU_ZERO(&set); CPU_SET(pr[(size_t) (t + (same ? nwalk : 0)) % pr.size()], &set);
            (void) pthread_setaffinity_np(rest[t]->native_handle(), sizeof(set), &set);
        }
        if (pinned) {
            pinned->clear();
            for (int t = 0; t < nwalk && t < (int) pw.size(); ++t) pinned->push_back(pw[(size_t) t]);
            for (int t = 0; t < nrest && t < (int) pr.size(); ++t) {
                const int cpu = pr[(size_t) (t + (same ? nwalk : 0)) % pr.size()];
                if (std::find(pinned->begin(), pinned->end(), cpu) == pinned->end()) pinned->push_back(cpu);
            }
        }
    } else {                               // no cache topology in sysfs: the whole node
        CPU_ZERO(&set);
        for (int k : cpus) CPU_SET(k, &set);
        for (int t = 0; t < nwalk; ++t) (void) pthread_setaffinity_np(walk[t]->native_handle(), sizeof(set), &set);
        for (int t = 0; t < nrest; ++t) (void) pthread_setaffinity_np(rest[t]->native_handle(), sizeof(set), &set);
    }
}

extern "C" {

#undef mgpu_config_defaults
static void config_defaults(struct mgpu_config *cfg);
void mgpu_config_defaults(struct mgpu_config *cfg) { config_defaults(cfg); }
// the host's struct may be an older, shorter one: nothing is written behind it, and the version it names is what mgpu_create checks
void mgpu_config_defaults_abi(struct mgpu_config *cfg, uint32_t struct_bytes, uint32_t abi_version) {
    struct mgpu_config full;
    config_defaults(&full);
    full.abi_version = abi_version;
    std::memcpy(cfg, &full, struct_bytes < sizeof(full) ? struct_bytes : sizeof(full));
}
static void config_defaults(struct mgpu_config *cfg) {
    std::memset(cfg, 0, sizeof(*cfg));
    cfg->device = 0;
    cfg->format = MGPU_FMT_UC8;
    cfg->nfix_crc = 1;                 // readsb.c:150
    cfg->fixDF = 1;                    // readsb.c:194
    cfg->preamble_threshold = 58;      // readsb.c:2268
    cfg->buf_samples = 131072;         // 256 KiB / 2 (readsb.c:228, 2212)
    cfg->trailing_samples = kTrailing; // readsb.c:288
    cfg->max_samples = 64ull * 131072;
    cfg->startup_time_ms = 0;
    cfg->abi_version = MGPU_ABI_VERSION;
}

uint32_t mgpu_abi_version(void) { return MGPU_ABI_VERSION; }

const char *mgpu_strerror(int code) {
    switch (code) {
        case MGPU_OK: return "ok";
        case MGPU_E_INVAL: return "invalid argument or state";
        case MGPU_E_NODEVICE: return "no usable HIP device (libmodes_gpu has no CPU fallback)";
        case MGPU_E_HIP: return "HIP runtime error";
        case MGPU_E_NOMEM: return "out of memory";
        case MGPU_E_OVERFLOW: return "device record pool overflow";
        case MGPU_E_CAPACITY: return "more samples than max_samples";
        case MGPU_E_EOF: return "stream already ended on a short buffer";
        default: return "unknown error";
    }
}

const char *mgpu_last_error(mgpu_ctx *ctx) { return ctx ? ctx->err.c_str() : ""; }

int mgpu_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

static int alloc_slot(mgpu_ctx *c, Slot &sl) {
    const uint64_t n = c->chunk_samples;
    const uint64_t mag_len = (n + kTile - 1) / kTile * kTile + kTile + kHalo + 64;
    HIPCHK(c, hipMalloc(&sl.d_mag, mag_len * sizeof(uint16_t)));
    HIPCHK(c, hipMemsetAsync(sl.d_mag, 0, mag_len * sizeof(uint16_t), c->stream));
    HIPCHK(c, hipMalloc(&sl.d_pool, c->cap_pool * sizeof(PhaseRec)));
    HIPCHK(c, hipMalloc(&sl.d_unit_first, (c->cap_units * (size_t) (kUnit / 2048) + 1) * sizeof(uint32_t)));   // one record chain per k_slice tile
    HIPCHK(c, hipMalloc(&sl.d_unit_count, (c->cap_units * (size_t) (kUnit / 2048) + 1) * sizeof(uint32_t)));
    HIPCHK(c, hipMalloc(&sl.d_unit_live, (c->cap_units + 2 + 3 * (c->cap_units / 4 + 2)) * sizeof(uint32_t)));   // per unit, then per count-pass workgroup: live records, the two class counts
    sl.class_bytes = (mag_len / 32 + 64 + kUnit / 32) * sizeof(uint32_t);      // (the count pass writes whole units: kUnit / 32 words each)
    HIPCHK(c, hipMalloc(&sl.d_class_final, sl.class_bytes));
    // the class planes and the scratch block are handed back zeroed by the kernels that consume them
    HIPCHK(c, hipMalloc(&sl.d_cand, (c->cap_units * (size_t) kUnit + 64) * sizeof(uint16_t)));
    HIPCHK(c, hipMalloc(&sl.d_cand_count, (c->cap_units * (size_t) (kUnit / kSweepTile) + 1) * sizeof(uint32_t)));
    HIPCHK(c, hipMalloc(&sl.d_dealer, (size_t) 2 * kDealerCounters * kDealerStride * sizeof(uint32_t)));   // (k_publish hands it back zeroed)
    HIPCHK(c, hipMemsetAsync(sl.d_dealer, 0, (size_t) 2 * kDealerCounters * kDealerStride * sizeof(uint32_t), c->stream));
    HIPCHK(c, hipMalloc(&sl.d_sweep_part, ((size_t) kSweepGridMax * 4) * sizeof(uint32_t)));
    {
        const size_t nb = c->cap_buffers, words = CNT_NUM + 1 + 4 * nb + kAcLists;   // ... + Mode A/C list counters
        sl.scratch_bytes = words * sizeof(unsigned long long);
        HIPCHK(c, hipMalloc(&sl.d_scratch, sl.scratch_bytes));
        HIPCHK(c, hipMemsetAsync(sl.d_scratch, 0, sl.scratch_bytes, c->stream));
        HIPCHK(c, hipHostMalloc(&sl.h_scratch, sl.scratch_bytes));
        sl.d_counters = sl.d_scratch;
        sl.d_pool_used = (uint32_t *) (sl.d_scratch + CNT_NUM);
        sl.d_sum_level = sl.d_scratch + CNT_NUM + 1;
        sl.d_sum_power = sl.d_sum_level + nb;
        sl.d_fsum_level = (double *) (sl.d_sum_power + nb);
        sl.d_fsum_power = sl.d_fsum_level + nb;
        sl.h_counters = sl.h_scratch;
        sl.h_sums = sl.h_scratch + CNT_NUM + 1;                 // level[nb] then power[nb]
        sl.h_fsums = (double *) (sl.h_scratch + CNT_NUM + 1 + 2 * nb);   // level[nb] then power[nb]
        if (c->cfg.format != MGPU_FMT_UC8) {
            HIPCHK(c, hipMalloc(&sl.d_fsx, 2 * nb * sizeof(double)));
            HIPCHK(c, hipHostMalloc(&sl.h_fsx, 2 * nb * sizeof(double)));
            HIPCHK(c, hipMemsetAsync(sl.d_fsx, 0, 2 * nb * sizeof(double), c->stream));
            sl.d_fsum_level = sl.d_fsx; sl.d_fsum_power = sl.d_fsx + nb; sl.h_fsums = sl.h_fsx;
        }
    }
    HIPCHK(c, hipMalloc(&sl.d_win, 8 * sizeof(unsigned long long)));
    HIPCHK(c, hipMalloc(&sl.d_win_part, kWinPartWords * sizeof(unsigned long long)));
    HIPCHK(c, hipMalloc(&sl.d_msg_pos, c->cap_msgs * sizeof(uint32_t)));
    HIPCHK(c, hipMalloc(&sl.d_msg_limit, c->cap_msgs * sizeof(uint32_t)));
    HIPCHK(c, hipMalloc(&sl.d_msg_len, c->cap_msgs * sizeof(uint16_t)));
    HIPCHK(c, hipMalloc(&sl.d_msg_skip, c->cap_msgs * sizeof(uint16_t)));
    HIPCHK(c, hipMalloc(&sl.d_msg_sig, c->cap_msgs * sizeof(unsigned long long)));
    {
        const size_t blob = c->cap_msgs * sizeof(Accepted) + c->cap_buffers * sizeof(BufferClock) + 64;
        HIPCHK(c, hipMalloc(&sl.d_blob, blob));
        HIPCHK(c, hipHostMalloc(&sl.h_blob, blob));
    }
    HIPCHK(c, hipMalloc(&sl.d_live, c->cap_pool * sizeof(PhaseRec)));
    HIPCHK(c, hipMalloc(&sl.d_live_sig, c->cap_pool * sizeof(unsigned long long)));
    HIPCHK(c, hipHostMalloc(&sl.h_live, c->cap_pool * sizeof(PhaseRec)));
    HIPCHK(c, hipHostMalloc(&sl.h_live_sig, c->cap_pool * sizeof(unsigned long long)));
    HIPCHK(c, hipEventCreateWithFlags(&sl.ev_window, hipEventDisableTiming));
    HIPCHK(c, hipEventCreateWithFlags(&sl.ev_scan, hipEventDisableTiming));
    HIPCHK(c, hipEventCreateWithFlags(&sl.ev_swept, hipEventDisableTiming));
    HIPCHK(c, hipEventCreateWithFlags(&sl.ev_conv, hipEventDisableTiming));
    HIPCHK(c, hipEventCreate(&sl.ev_sweep0));
    HIPCHK(c, hipEventCreateWithFlags(&sl.ev_h2d, hipEventDisableTiming));
    HIPCHK(c, hipEventCreateWithFlags(&sl.ev_pre, hipEventDisableTiming));
    HIPCHK(c, hipEventCreateWithFlags(&sl.ev_fsum, hipEventDisableTiming));
    HIPCHK(c, hipEventCreateWithFlags(&sl.ev_convdone, hipEventDisableTiming));
    HIPCHK(c, hipHostMalloc(&sl.h_win, 8 * sizeof(unsigned long long)));
    HIPCHK(c, hipHostMalloc(&sl.h_sig, c->cap_msgs * sizeof(unsigned long long)));
    HIPCHK(c, hipHostMalloc(&sl.h_msg_pos, c->cap_msgs * sizeof(uint32_t)));
    HIPCHK(c, hipHostMalloc(&sl.h_msg_limit, c->cap_msgs * sizeof(uint32_t)));
    HIPCHK(c, hipHostMalloc(&sl.h_msg_len, c->cap_msgs * sizeof(uint16_t)));
    HIPCHK(c, hipHostMalloc(&sl.h_msg_skip, c->cap_msgs * sizeof(uint16_t)));
    if (c->cfg.mode_ac) {
        HIPCHK(c, hipMalloc(&sl.d_ac_noise, (c->cap_buffers + 1) * sizeof(uint32_t)));
        HIPCHK(c, hipHostMalloc(&sl.h_ac, c->cap_ac * sizeof(AcCand)));
    }
    for (auto &e : sl.ev) HIPCHK(c, hipEventCreate(&e));
    HIPCHK(c, hipEventCreateWithFlags(&sl.ev_done, hipEventDisableTiming));
    return MGPU_OK;
}

static void free_slot(Slot &sl) {
    if (sl.h_blob) (void) hipHostFree(sl.h_blob);
    if (sl.h_live_win) (void) hipHostFree(sl.h_live_win);
    if (sl.h_fsx) (void) hipHostFree(sl.h_fsx);
    if (sl.d_fsx) (void) hipFree(sl.d_fsx);
    void *dev[] = {sl.d_live_win, sl.d_dealer, sl.d_blob, sl.d_live, sl.d_live_sig, sl.d_mag, sl.d_pool, sl.d_scratch, sl.d_unit_first, sl.d_unit_count, sl.d_unit_live,
                   sl.d_class_final, sl.d_cand, sl.d_cand_count, sl.d_sweep_part,
                   sl.d_win, sl.d_win_part, sl.d_msg_pos,
                   sl.d_msg_limit, sl.d_msg_len, sl.d_msg_skip, sl.d_msg_sig, sl.d_wk_in, sl.d_wk_acc, sl.d_wk_sig};
    for (void *p : dev)
        if (p) (void) hipFree(p);
    if (sl.ev_window) (void) hipEventDestroy(sl.ev_window);
    if (sl.ev_scan) (void) hipEventDestroy(sl.ev_scan);
    if (sl.ev_swept) (void) hipEventDestroy(sl.ev_swept);
    if (sl.ev_conv) (void) hipEventDestroy(sl.ev_conv);
    if (sl.ev_sweep0) (void) hipEventDestroy(sl.ev_sweep0);
    if (sl.d_ac_noise) (void) hipFree(sl.d_ac_noise);
    if (sl.h_ac) (void) hipHostFree(sl.h_ac);
    if (sl.ev_h2d) (void) hipEventDestroy(sl.ev_h2d);
    if (sl.ev_pre) (void) hipEventDestroy(sl.ev_pre);
    if (sl.ev_fsum) (void) hipEventDestroy(sl.ev_fsum);
    if (sl.ev_convdone) (void) hipEventDestroy(sl.ev_convdone);
    if (sl.ev_done) (void) hipEventDestroy(sl.ev_done);
    void *host[] = {sl.h_live, sl.h_live_sig, sl.h_scratch, sl.h_win, sl.h_sig, sl.h_msg_pos,
                    sl.h_msg_limit, sl.h_msg_len, sl.h_msg_skip};
    for (void *p : host)
        if (p) (void) hipHostFree(p);
    for (auto &e : sl.ev)
        if (e) (void) hipEventDestroy(e);
}

static int alloc_all(mgpu_ctx *c) {
    const mgpu_config &cfg = c->cfg;
    const uint64_t n = cfg.max_samples;
    c->cap_samples = n;
    // pipeline chunk: a whole number of 131072-sample buffers (cfg.chunk_buffers)
    uint64_t chunk_buffers = cfg.chunk_buffers ? cfg.chunk_buffers : 1024;   // (round 4: 512 -> 1024, 300 -> 325 Gsamples/s once the builder kept up; profiles/r04_chunk_buffers.txt)
    c->chunk_samples = chunk_buffers * cfg.buf_samples;
    const uint64_t nmax_buffers = (n + cfg.buf_samples - 1) / cfg.buf_samples;
    if (c->chunk_samples > nmax_buffers * cfg.buf_samples) c->chunk_samples = nmax_buffers * cfg.buf_samples;
    const uint64_t cs = c->chunk_samples;
    c->cap_units = (cs + kUnit - 1) / kUnit;
    c->cap_buffers = (cs + cfg.buf_samples - 1) / cfg.buf_samples + 1;
    // default pool: 1 record per 16 samples (8x the density of busy synthetic traffic) + what the
    // workgroups reserve but may leave unused (one chunk each)
    const uint64_t reserve = 1024ull * (c->cap_units < (uint64_t) kSweepMaxWaves ? c->cap_units : (uint64_t) kSweepMaxWaves);
    uint64_t pool = cfg.record_pool_records ? cfg.record_pool_records : cs / 16 + 65536;
    c->cap_pool = pool + reserve;
    if (c->cap_pool > 0xFFFFFFF0ull) c->cap_pool = 0xFFFFFFF0ull;
    c->cap_msgs = cfg.max_messages ? cfg.max_messages : cs / 64 + 65536;
    c->cap_ac = cs / 256 + 65536;          // Mode A/C candidates per chunk (a reply lasts 49 samples)
    const size_t bps = cfg.format == MGPU_FMT_UC8 ? 2 : 4;

    HIPCHK(c, hipMalloc(&c->d_iq, n * bps + 64));
    c->ev_iq_read.assign((size_t) ((n + cs - 1) / cs) + 1, nullptr);
    c->iq_region_used.assign(c->ev_iq_read.size(), 0);
    for (auto &e : c->ev_iq_read) HIPCHK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    HIPCHK(c, hipMalloc(&c->d_win, 8 * sizeof(unsigned long long)));
    HIPCHK(c, hipHostMalloc(&c->h_win, 8 * sizeof(unsigned long long)));
    HIPCHK(c, hipMalloc(&c->d_adder_bitmap, (1u << 24) / 8));
    HIPCHK(c, hipMemsetAsync(c->d_adder_bitmap, 0, (1u << 24) / 8, c->stream));
    for (auto &sl : c->slot) {
        int rc = alloc_slot(c, sl);
        if (rc != MGPU_OK) return rc;
    }

    if (cfg.format != MGPU_FMT_UC8)
        for (auto &r : c->fsum_ring) {
            HIPCHK(c, hipHostMalloc(&r.h, 2 * c->cap_buffers * sizeof(double)));
#if MGPU_EXPERIMENTS
            HIPCHK(c, hipMalloc(&r.d, 2 * c->cap_buffers * sizeof(double)));
            HIPCHK(c, hipMalloc(&r.scratch, fsum_wide_scratch_bytes(c->chunk_samples, cfg.buf_samples)));
#endif
            HIPCHK(c, hipEventCreateWithFlags(&r.ev, hipEventDisableTiming));
        }
    for (auto &j : c->job) {
        HIPCHK(c, hipHostMalloc(&j.h_msig, c->cap_msgs * sizeof(unsigned long long)));
        HIPCHK(c, hipEventCreateWithFlags(&j.ev_copied, hipEventDisableTiming));
    }
#if MGPU_EXPERIMENTS
    if (c->device_walk) {
        WalkBuffers &w = c->wk;
        const size_t nb = c->cap_buffers + 1;
        c->wk_acc_cap = cfg.buf_samples / 112 + 16;          // an accepted frame hides the next 112 positions at least
        HIPCHK(c, hipMalloc(&w.bit_active, (1u << 24) / 8));
        HIPCHK(c, hipMalloc(&w.bit_inactive, (1u << 24) / 8));
        HIPCHK(c, hipMemsetAsync(w.bit_active, 0, (1u << 24) / 8, c->stream));
        HIPCHK(c, hipMemsetAsync(w.bit_inactive, 0, (1u << 24) / 8, c->stream));
        for (int t = 0; t < 2; ++t) {
            HIPCHK(c, hipMalloc(&w.first[t], sizeof(uint32_t) << 24));
            HIPCHK(c, hipMemsetAsync(w.first[t], 0xff, sizeof(uint32_t) << 24, c->stream));
            HIPCHK(c, hipMalloc(&w.touched[t], kWkTouchedCap * sizeof(uint32_t)));
        }
        HIPCHK(c, hipMalloc(&w.state, sizeof(WalkState)));
        HIPCHK(c, hipMemsetAsync(w.state, 0, sizeof(WalkState), c->stream));
        HIPCHK(c, hipMalloc(&w.rec_lo, nb * sizeof(uint32_t)));
        HIPCHK(c, hipMalloc(&w.nacc, nb * sizeof(uint32_t)));
        HIPCHK(c, hipMalloc(&w.nadds, nb * sizeof(uint32_t)));
        HIPCHK(c, hipMalloc(&w.counts, nb * kWkCounts * sizeof(uint32_t)));
        HIPCHK(c, hipMalloc(&w.end_clock, nb * sizeof(long long)));
        HIPCHK(c, hipMalloc(&w.acc, nb * c->wk_acc_cap * 2 * sizeof(uint32_t)));
        HIPCHK(c, hipMalloc(&w.adds, nb * c->wk_acc_cap * sizeof(uint32_t)));
        HIPCHK(c, hipMalloc(&w.offs, nb * sizeof(uint32_t)));
        HIPCHK(c, hipMalloc(&w.aoffs, nb * sizeof(uint32_t)));
        c->wk_in_cap = walk_input_bytes((uint32_t) c->cap_buffers, 1u << 17, 1u << 17);
        HIPCHK(c, hipHostMalloc(&c->h_wk_in, c->wk_in_cap));
        for (auto &sl : c->slot) {
            HIPCHK(c, hipMalloc(&sl.d_wk_in, c->wk_in_cap));
            HIPCHK(c, hipMalloc(&sl.d_wk_acc, c->cap_msgs * 12));
            HIPCHK(c, hipMalloc(&sl.d_wk_sig, c->cap_msgs * sizeof(unsigned long long)));
        }
        const WalkState st0 = {kWkNoFlip, 0, 0, 0, {0, 0}, 0, 0};
        HIPCHK(c, hipMemcpyAsync(w.state, &st0, sizeof(st0), hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        HIPCHK(c, hipHostMalloc(&c->h_wk_sum, walk_summary_bytes((uint32_t) c->cap_buffers, c->wk_acc_cap)));
        if (c->device_walk == 1) {
            HIPCHK(c, hipMalloc(&c->d_wk_msgs, c->cap_msgs * sizeof(mgpu_msg)));
            for (auto &j : c->job) HIPCHK(c, hipHostMalloc(&j.h_msgs, c->cap_msgs * sizeof(mgpu_msg)));
        }
        HIPCHK(c, hipEventCreateWithFlags(&c->ev_wk, hipEventDisableTiming));
    }

#endif
    // constant tables
    const CrcTables &crc = crc_tables();
    HIPCHK(c, hipMalloc(&c->d_bit_syndrome, 112 * sizeof(uint32_t)));
    HIPCHK(c, hipMemcpy(c->d_bit_syndrome, crc.bit_syndrome, 112 * sizeof(uint32_t), hipMemcpyHostToDevice));
    const std::vector<uint32_t> gs = build_group_syndromes();
    HIPCHK(c, hipMalloc(&c->d_group_syndrome, gs.size() * sizeof(uint32_t)));
    HIPCHK(c, hipMemcpy(c->d_group_syndrome, gs.data(), gs.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    const ParityMasks pm = build_parity_masks();
    HIPCHK(c, hipMalloc(&c->d_parity, sizeof(pm)));
    HIPCHK(c, hipMemcpy(c->d_parity, &pm, sizeof(pm), hipMemcpyHostToDevice));
    c->tab_long = build_syndrome_table(112, cfg.nfix_crc);
    c->tab_short = build_syndrome_table(56, cfg.nfix_crc);
    c->n_long = (int) c->tab_long.size();
    c->n_short = (int) c->tab_short.size();
    if (c->n_long > 4096 || c->n_short > 4096) { c->err = "syndrome table too large for the wave search"; return MGPU_E_INVAL; }
    const std::vector<uint64_t> pl = pack_syndrome_table(c->tab_long), ps = pack_syndrome_table(c->tab_short);
    HIPCHK(c, hipMalloc(&c->d_tab_long, (pl.size() + 1) * sizeof(uint64_t)));
    HIPCHK(c, hipMalloc(&c->d_tab_short, (ps.size() + 1) * sizeof(uint64_t)));
    if (!pl.empty()) HIPCHK(c, hipMemcpy(c->d_tab_long, pl.data(), pl.size() * sizeof(uint64_t), hipMemcpyHostToDevice));
    if (!ps.empty()) HIPCHK(c, hipMemcpy(c->d_tab_short, ps.data(), ps.size() * sizeof(uint64_t), hipMemcpyHostToDevice));
    const std::vector<uint16_t> folded = uc8_folded_table();
    HIPCHK(c, hipMalloc(&c->d_uc8_folded, folded.size() * sizeof(uint16_t)));
    HIPCHK(c, hipMemcpy(c->d_uc8_folded, folded.data(), folded.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return MGPU_OK;
}

int mgpu_create(const struct mgpu_config *cfg, mgpu_ctx **out) {
    if (!cfg || !out) return MGPU_E_INVAL;
    *out = nullptr;
    if (cfg->trailing_samples != (uint32_t) kTrailing || cfg->buf_samples == 0 || cfg->buf_samples % kTile != 0 ||
        cfg->max_samples == 0 || cfg->format < 0 || cfg->format > 2 || cfg->nfix_crc < 0 || cfg->nfix_crc > 2 ||
        cfg->filter_clock > MGPU_FILTER_CLOCK_EXTERNAL ||
        cfg->abi_version != MGPU_ABI_VERSION ||           // the host's header is another version's (or it did not call mgpu_config_defaults)
        cfg->chunk_buffers > 16384u ||                    // (0 = the default; 16384 buffers = 2^31 samples: positions are 32 bits)
        // k_sweep's threshold tests run in 32-bit accumulators on 16-bit coefficients: thr * (five samples) + 32 * (four samples) must
        // stay below 2^31 — thr <= 6000 — and a threshold below 1 accepts every position.  The reference's own range is 40..400
        // (demod_2400.h:28-34, clamped at readsb.c:1473).
        cfg->preamble_threshold < 1 || cfg->preamble_threshold > 4095)
        return MGPU_E_INVAL;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || cfg->device >= ndev) return MGPU_E_NODEVICE;
    mgpu_ctx *c = new (std::nothrow) mgpu_ctx();
    if (!c) return MGPU_E_NOMEM;
    c->cfg = *cfg;
    if (hipSetDevice(cfg->device) != hipSuccess) { delete c; return MGPU_E_NODEVICE; }
    // the second stream (what follows a chunk's walk; the SC16 formats' float sums) at the lowest priority the device offers: its
    // kernels fill what the main stream leaves, they are not to take its slots
    int prio_least = 0, prio_greatest = 0;
    (void) hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
#if MGPU_EXPERIMENTS
    if (const char *e = getenv("MGPU_S2_PRIORITY")) prio_least = atoi(e) > 0 ? prio_greatest : 0;   // experiment: the second stream at normal (0) / highest (1) priority
#endif
    // The side streams — the second stream (what follows a walk: k_stage_in, k_window_stats, k_build_messages), the fetcher's record
    // copies (the runtime's blit kernel) and, for the SC16 formats, the float-sum chain (k_fsum_sc16: one wave per buffer, ~0.9 ms of
    // dependent block steps per chunk) — are created through hipExtStreamCreateWithCUMask with EVERY CU enabled.  What that buys is not
    // a place but a queue: such a stream has a hardware queue of its own, where the runtime's ordinary streams share a small pool of
    // them (four by default; bench.py asks for eight, GPU_MAX_HW_QUEUES, which a library cannot count on) and a kernel of one waits
    // behind another's.  Measured (profiles/r06_stream_queues.txt, on the pool of eight): UC8 headline 466-486 Gsamples/s against
    // 426-478 with ordinary streams of any priority; SC16Q11 --aggressive 273-285 against 224-245 with the chain on an ordinary stream.
    // Round 6's first form asked for "every 8th CU" (every 4th for the chain) and believed the side work confined there.  It is not:
    // mask bit i is CU i / 8 of XCC i % 8 (tools/micro/cu_mask_map.hip), a stride of 8 selects all of XCC 0, and an XCC whose share of
    // the mask is empty runs the queue's workgroups on all of its CUs — those masks were the whole device, and what they gained was
    // the queue.  Real confinement (n CUs of every XCC) LOSES: the side streams on 8 / 4 / 2 CUs per XCC 431-454 / 402-422 / 306-312,
    // the chain on 16 / 8 / 4 per XCC 251-258 / 219-224 / 145; a stride of 3 (10-11 CUs per XCC) 225-229.
    const int n_cus = [&] { hipDeviceProp_t prop; return hipGetDeviceProperties(&prop, cfg->device) == hipSuccess && prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256; }();
    // mask: every k-th bit from `off` (k = 1: every CU); perxcc > 0: CUs [first, first + perxcc) of each of the 8 XCCs instead
    auto build_mask = [&](uint32_t *mask, int k, int off, int perxcc, int first) -> bool {
        if (n_cus > 1024 || (perxcc <= 0 && (k < 1 || k > 128 || n_cus < 2 * k))) return false;
        if (perxcc > 0) { for (int b = 8 * first; b < 8 * (first + perxcc) && b < n_cus; ++b) mask[b >> 5] |= 1u << (b & 31); }
        else for (int cu = off % k; cu < n_cus; cu += k) mask[cu >> 5] |= 1u << (cu & 31);
        return true;
    };
    const uint32_t mask_words = (uint32_t) ((n_cus + 31) / 32);
    bool masked = false;
    {
        int k = 1, perxcc = 0, perxcc_first = 0;
#if MGPU_EXPERIMENTS
        if (const char *e = getenv("MGPU_CU_MASK_STRIDE")) k = atoi(e);      // 0: ordinary streams (A/B); k >= 2: every k-th bit
        if (const char *e = getenv("MGPU_CU_MASK_PERXCC")) sscanf(e, "%d,%d", &perxcc, &perxcc_first);
#endif
        if (build_mask(c->cu_mask, k, 0, perxcc, perxcc_first)) {
            c->cu_mask_words = mask_words;
            int which = 3;                                                   // experiment: 1 = only the second stream, 2 = only the record copies' stream
#if MGPU_EXPERIMENTS
            if (const char *e = getenv("MGPU_OWN_QUEUES")) which = atoi(e) & 3;
#endif
            masked = (!(which & 1) || hipExtStreamCreateWithCUMask(&c->stream2, mask_words, c->cu_mask) == hipSuccess) &&
                     (!(which & 2) || hipExtStreamCreateWithCUMask(&c->stream_d2h, mask_words, c->cu_mask) == hipSuccess);
            if (!masked) {
                (void) hipGetLastError();
                if (c->stream2) (void) hipStreamDestroy(c->stream2);
                if (c->stream_d2h) (void) hipStreamDestroy(c->stream_d2h);
                c->stream2 = c->stream_d2h = nullptr;
            }
        }
    }
    bool fsum_masked = false;
    int fsum_prio = -1;
    if (cfg->format != MGPU_FMT_UC8) {
        int k = 1, off = 0, perxcc = 0, perxcc_first = 0;
#if MGPU_EXPERIMENTS
        if (const char *e = getenv("MGPU_FSUM_CU_STRIDE")) { k = 0; off = 1; sscanf(e, "%d,%d", &k, &off); }      // k[,off]; 0: an ordinary stream (A/B)
        if (const char *e = getenv("MGPU_FSUM_CU_PERXCC")) sscanf(e, "%d,%d", &perxcc, &perxcc_first);              // n[,first]: CUs [first, first + n) of every XCC
        if (const char *e = getenv("MGPU_FSUM_PRIORITY")) fsum_prio = atoi(e);                                      // the ordinary stream's priority: -1 least (rounds 3-6), 0 normal, 1 greatest
#endif
        uint32_t m[32] = {0};
        if (build_mask(m, k, off, perxcc, perxcc_first)) {
            fsum_masked = hipExtStreamCreateWithCUMask(&c->stream_f, mask_words, m) == hipSuccess;
            if (!fsum_masked) { (void) hipGetLastError(); c->stream_f = nullptr; }
            // (Measured and dropped: a SECOND such stream, the chunks' chains on the two in turn — with its own queue the chain is what
            // bounds SC16Q11 --aggressive: two chunks' chains are 1.87 ms per segment of 537 M samples, the builder waits 0.45-0.8 ms per
            // segment for the sums, the main stream's kernels take 1.74.  Two chains at once each take twice as long and are in the main
            // kernels' way twice: 240-243 against 268-274 Gsamples/s, tools/ab/ab_fsum2.sh, profiles/r06_stream_queues.txt (6).)
        }
    }
#if MGPU_EXPERIMENTS
    // The main stream (1), and the upload stream (2), through the mask API too: measured — the main stream with a queue of its own is
    // no faster alone and 1-8 % slower in three interleaved pairs of the plain and of the aggregator path (profiles/r06_stream_queues.txt
    // (4)); every further queue of their own costs a fan-in's contexts (four streams 14.0 against 19.5 Gsamples/s, profiles/r06_fanin.txt).
    if (const char *oq = getenv("MGPU_MAIN_OWN_QUEUE")) {
        uint32_t m[32] = {0};
        if (atoi(oq) >= 1 && build_mask(m, 1, 0, 0, 0)) {
            if (hipExtStreamCreateWithCUMask(&c->stream, mask_words, m) != hipSuccess) { (void) hipGetLastError(); c->stream = nullptr; }
            if (atoi(oq) >= 2 && hipExtStreamCreateWithCUMask(&c->stream_w, mask_words, m) != hipSuccess) { (void) hipGetLastError(); c->stream_w = nullptr; }
        }
    }
#endif
    if ((!c->stream && hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) ||
        (!c->stream2 && hipStreamCreateWithPriority(&c->stream2, hipStreamNonBlocking, prio_least) != hipSuccess) ||
        (!c->stream_w && hipStreamCreateWithFlags(&c->stream_w, hipStreamNonBlocking) != hipSuccess) ||
        (!c->stream_d2h && hipStreamCreateWithFlags(&c->stream_d2h, hipStreamNonBlocking) != hipSuccess) ||
        hipStreamCreateWithFlags(&c->stream_c, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&c->stream_aux, hipStreamNonBlocking) != hipSuccess ||
        (cfg->format != MGPU_FMT_UC8 && !fsum_masked && hipStreamCreateWithPriority(&c->stream_f, hipStreamNonBlocking, fsum_prio < 0 ? prio_least : fsum_prio > 0 ? prio_greatest : 0) != hipSuccess)) { mgpu_destroy(c); return MGPU_E_HIP; }
    // valid_df_*_bitset, init_bitsets() demod_2400.c:112-128 (ENABLE_DF24 off, readsb.h:303)
    c->valid_short = (1u << 0) | (1u << 4) | (1u << 5) | (1u << 11);
    c->valid_long = (1u << 16) | (1u << 17) | (1u << 18) | (1u << 20) | (1u << 21);
    if (cfg->fixDF && cfg->nfix_crc)
        for (int b = 0; b < 5; ++b) c->valid_long |= 1u << (17 ^ (1 << b));
#if MGPU_EXPERIMENTS   // the ordered walk on the device (kernels/walk.inc): the experiments build only — DESIGN.md §3 says why the host owns the walk
    if (const char *e = getenv("MGPU_DEVICE_WALK")) c->device_walk = !strcmp(e, "check") ? 2 : atoi(e) != 0;
    c->wk_serial_only = getenv("MGPU_DBG_WK_SERIAL") != nullptr;
    if (c->device_walk) {
        int lo = 0, hi = 0;
        (void) hipDeviceGetStreamPriorityRange(&lo, &hi);
        if (hipStreamCreateWithPriority(&c->stream_wk, hipStreamNonBlocking, hi) != hipSuccess) { mgpu_destroy(c); return MGPU_E_HIP; }
    }
#endif
    c->s_post = c->stream2;
    int rc;
    {
        NearDevice near(cfg->device);
        rc = alloc_all(c);
    }
    if (rc != MGPU_OK) {
        std::fprintf(stderr, "mgpu_create: %s (%s)\n", mgpu_strerror(rc), c->err.c_str());
        mgpu_destroy(c);
        return rc;
    }
    c->resolver.reset(cfg->startup_time_ms, (int) cfg->filter_clock);
#if MGPU_EXPERIMENTS   // debug / experiment switches: the experiments build only; the product reads MGPU_NO_AFFINITY and nothing else
    c->dbg_print = getenv("MGPU_DEBUG_PRINT") != nullptr;
    if (const char *e = getenv("MGPU_DEBUG_STAGE")) c->dbg_stage = atoi(e);   // (k_slice with one part left out: tools/slice_stages.sh)
    if (const char *e = getenv("MGPU_TIMING_EVERY")) { const int v = atoi(e); if (v >= 1) c->timing_every = v; }
    if (const char *e = getenv("MGPU_DUMP_DIR")) { c->dump_dir = e; c->sig_late = false; }   // (the dump holds per-record signal powers)
    c->fsum_wide = getenv("MGPU_FSUM_WIDE") != nullptr;
    if (const char *e = getenv("MGPU_WRITE_BESIDE")) {     // 1: write pass + k_publish beside the next chunk's sweep; 2: the count pass too; 3 / 4: the same on a mask-API stream (a queue of its own)
        const int v = atoi(e);
        c->post_beside = v == 2 || v == 4 ? 2 : v ? 1 : 0;
        const bool m = v >= 3 && c->cu_mask_words;
        if (v && (m ? hipExtStreamCreateWithCUMask(&c->stream_pw, c->cu_mask_words, c->cu_mask) : hipStreamCreateWithFlags(&c->stream_pw, hipStreamNonBlocking)) != hipSuccess) { mgpu_destroy(c); return MGPU_E_HIP; }
    }
    if (const char *e = getenv("MGPU_CONVERT_OLD")) c->convert_variant = atoi(e) ? 1 : 0;          // A/B: the round-1..5 UC8 converter
    if (const char *e = getenv("MGPU_S2_HOLD")) c->s2_hold = atoi(e);
    if (const char *e = getenv("MGPU_D2H_HOLD")) c->d2h_hold = atoi(e);
    if (const char *e = getenv("MGPU_SWEEP_FUSED")) c->sweep_fused = atoi(e);                      // A/B: converter and sweep in one kernel (1, the product) or two
    if (const char *e = getenv("MGPU_CONV_SIDE")) c->conv_side = atoi(e);                          // A/B: the converter beside k_slice
    if (const char *e = getenv("MGPU_CONV_SIDE_BLOCKS")) c->conv_side_blocks = (unsigned) atoi(e);
    if (const char *e = getenv("MGPU_SLICE_BLOCKS")) c->slice_blocks_cap = (unsigned) atoi(e);
    if (const char *e = getenv("MGPU_PRESCREEN_VARIANT")) c->prescreen_variant = (uint32_t) atoi(e);   // (measured, r04k: no faster than the chain per buffer, twice its HBM traffic)
#endif
    c->device_slot = take_device_slot(cfg->device);
    // the first context of a device has two L3 groups to itself (bind_near_device): a walk team of 8 and a builder team of 6;
    // further contexts of the device share one group: 4 + 3 as before
    if (c->device_slot == 0 && cfg->streams_on_device <= 1 && !getenv("MGPU_NO_AFFINITY")) { c->walk_threads = 8; c->build_threads = 6; }
    // with the walk on the device the walker only waits for the GPU and replays the adds; the builder copies and sums
    if (c->device_walk == 1) { c->walk_threads = 1; c->build_threads = 2; }
#if MGPU_EXPERIMENTS   // (8 + 6 is the best of 6..16 + 6..8; tools/stress.py varies them)
    if (const char *e = getenv("MGPU_WALK_THREADS")) { const int v = atoi(e); if (v >= 1 && v <= 64) c->walk_threads = v; }
    if (const char *e = getenv("MGPU_WALK_RANGES")) { const int v = atoi(e); if (v >= 2 && v <= 64) c->walk_ranges = v; }
    if (const char *e = getenv("MGPU_BUILD_THREADS")) { const int v = atoi(e); if (v >= 1 && v <= 64) c->build_threads = v; }
#endif
    c->fetcher = std::thread(fetcher_main, c);
    c->worker = std::thread(worker_main, c);
    c->builder = std::thread(builder_main, c);
    c->walk_team.start(c->walk_threads - 1, &c->hot);
    c->build_team.start(c->build_threads - 1, &c->hot);
    {
        std::vector<std::thread *> tw = {&c->worker}, tr = {&c->builder, &c->fetcher};
        for (auto &t : c->walk_team.threads) tw.push_back(&t);
        for (auto &t : c->build_team.threads) tr.push_back(&t);
        bind_near_device(tw.data(), (int) tw.size(), tr.data(), (int) tr.size(), cfg->device, c->device_slot, &c->host_cpus);
    }
    *out = c;
    return MGPU_OK;
}

void mgpu_destroy(mgpu_ctx *c) {
    if (!c) return;
    if (c->fetcher.joinable() || c->worker.joinable() || c->builder.joinable()) {
        { std::lock_guard<std::mutex> lk(c->mu); c->stop = true; }
        c->cv.notify_all();
        if (c->fetcher.joinable()) c->fetcher.join();
        if (c->worker.joinable()) c->worker.join();
        if (c->builder.joinable()) c->builder.join();
    }
    c->walk_team.stop();
    c->build_team.stop();
    if (c->device_slot >= 0) release_device_slot(c->cfg.device, c->device_slot);
    (void) hipSetDevice(c->cfg.device);
    if (c->stream) (void) hipStreamSynchronize(c->stream);
    if (c->stream2) (void) hipStreamSynchronize(c->stream2);
    if (c->stream_w) (void) hipStreamSynchronize(c->stream_w);
    if (c->stream_pw) (void) hipStreamSynchronize(c->stream_pw);
    if (c->stream_f) (void) hipStreamSynchronize(c->stream_f);
    if (c->stream_c) (void) hipStreamSynchronize(c->stream_c);
    if (c->stream_aux) (void) hipStreamSynchronize(c->stream_aux);
    if (c->stream_wk) (void) hipStreamSynchronize(c->stream_wk);
    for (auto &sl : c->slot) free_slot(sl);
    for (auto &f : c->feed) {
        if (f.d_msgs) (void) hipFree(f.d_msgs);
        if (f.ev_built) (void) hipEventDestroy(f.ev_built);
    }
    if (c->h_win) (void) hipHostFree(c->h_win);
    if (c->h_wk_in) (void) hipHostFree(c->h_wk_in);
    if (c->h_wk_sum) (void) hipHostFree(c->h_wk_sum);
    if (c->ev_wk) (void) hipEventDestroy(c->ev_wk);
    for (auto &r : c->fsum_ring) {
        if (r.d) (void) hipFree(r.d);
        if (r.scratch) (void) hipFree(r.scratch);
        if (r.h) (void) hipHostFree(r.h);
        if (r.ev) (void) hipEventDestroy(r.ev);
    }
    for (auto &j : c->job) {
        if (j.h_msgs) (void) hipHostFree(j.h_msgs);
        if (j.h_msig) (void) hipHostFree(j.h_msig);
        if (j.ev_copied) (void) hipEventDestroy(j.ev_copied);
    }
    {
        const WalkBuffers &w = c->wk;
        void *wkp[] = {w.bit_active, w.bit_inactive, w.first[0], w.first[1], w.touched[0], w.touched[1], w.state, w.rec_lo, w.nacc, w.nadds, w.counts,
                       w.end_clock, w.acc, w.adds, w.offs, w.aoffs, c->d_wk_msgs};
        for (void *p : wkp)
            if (p) (void) hipFree(p);
    }
    for (hipEvent_t e : c->ev_iq_read)
        if (e) (void) hipEventDestroy(e);
    void *dev[] = {c->d_deferred, c->d_gate_table, c->d_gate_scratch, c->d_gate_verdict, c->d_roll_tan, c->d_fields, c->d_beast_off, c->d_beast_len, c->d_beast_in, c->d_beast_out, c->d_beast_blocks, c->d_beast_total, c->d_hist, c->d_hist_iq, c->d_hist_sums, c->d_iq, c->d_win, c->d_adder_bitmap, c->d_bit_syndrome, c->d_group_syndrome, c->d_parity,
                   c->d_tab_long, c->d_tab_short, c->d_uc8_folded};
    for (void *p : dev)
        if (p) (void) hipFree(p);
    if (c->stream) (void) hipStreamDestroy(c->stream);
    if (c->stream2) (void) hipStreamDestroy(c->stream2);
    if (c->stream_w) (void) hipStreamDestroy(c->stream_w);
    if (c->stream_pw) (void) hipStreamDestroy(c->stream_pw);
    if (c->stream_d2h) (void) hipStreamDestroy(c->stream_d2h);
    if (c->stream_f) (void) hipStreamDestroy(c->stream_f);
    if (c->stream_c) (void) hipStreamDestroy(c->stream_c);
    if (c->stream_aux) (void) hipStreamDestroy(c->stream_aux);
    if (c->stream_wk) (void) hipStreamDestroy(c->stream_wk);
    delete c;
}

static int drain(mgpu_ctx *c);

int mgpu_reset(mgpu_ctx *c) {
    if (!c) return MGPU_E_INVAL;
    (void) drain(c);                         // deferred feeds still in flight land first (their results are discarded)
    HIPCHK(c, hipSetDevice(c->cfg.device));
    c->resolver.reset(c->cfg.startup_time_ms, (int) c->cfg.filter_clock);
    c->pending.clear();
    for (auto &f : c->feed) { f.msgs.clear(); f.jobs_total = f.jobs_built = 0; f.closed = false; }
    c->feed_head = c->feed_tail = 0;
    std::memset(&c->counters, 0, sizeof(c->counters));
    std::memset(&c->timing, 0, sizeof(c->timing));
    c->stream_pos = 0;
    c->eof = false;
    c->tail_src = nullptr;
    c->worker_rc = MGPU_OK;
    c->shard_mode = 0;
    c->shard_packets.clear();
    c->shard_est.clear(); c->shard_est_pos.clear(); c->shard_est_off.clear();
    c->resolver.set_schedule(nullptr, 0);
    c->resolver.log_end_clocks(nullptr);
    c->shard_noise_on = c->shard_stream = false;
    HIPCHK(c, hipMemsetAsync(c->d_adder_bitmap, 0, (1u << 24) / 8, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return MGPU_OK;
}

// ---- GPU half of a chunk: enqueue convert -> sweep/slice -> pre-screen on the main stream ----------
// `iq` = device pointer to the chunk's IQ samples (ignored when sl.have_mag: sl.d_mag already holds
// the magnitudes of one struct mag_buf).
// (Measured and dropped: the NEXT chunk's converter enqueued between a chunk's slicer and its post-sweep kernels, so that the
// fetcher's record copies — which start when k_publish has run and slow whatever streams memory at that moment by 30-50 us —
// would meet k_sweep instead of the converter: k_sweep then took 68 us instead of 37 and the step 2.85 ms instead of 2.52.)
// (Measured and dropped, round 5: the converters on a stream of their own, chunk N + 1's held behind chunk N's k_sweep so that it runs
// beside k_slice — issue-bound — and the post-sweep kernels: 305-325 against 363-367 Gsamples/s (gpurun r05z).  k_slice's persistent
// grid holds every CU's registers and LDS until it ends, so the converter really runs beside the post-sweep kernels, whose dependent
// round trips stretch from 0.25 to 0.5 ms per step next to a kernel that saturates the memory system, as with MGPU_WRITE_BESIDE.)
// HIP events with timing cost ~5 us of idle stream each (the next kernel waits for the marker): only every
// `timing_every`-th chunk carries the stage events (sl.timed); the others record the completion event alone.
// k_fsum_sc16 of the slot's chunk on the second stream, behind `after` (an event of the main stream)
static int enqueue_fsum(mgpu_ctx *c, Slot &sl, const uint8_t *iq, hipEvent_t after) {
    const mgpu_config &cfg = c->cfg;
    sl.fsum_idx = (int) (sl.seq % mgpu_ctx::kFsumRing);
    mgpu_ctx::FsumRing &r = c->fsum_ring[sl.fsum_idx];
    const size_t nb = c->cap_buffers;
    HIPCHK(c, hipStreamWaitEvent(c->stream_f, after, 0));
    // The chain's waves store their results straight into the ring entry's page-locked host memory (one double per buffer).  A
    // hipMemcpyAsync behind the kernel goes to the copy engine as a packet that waits for the kernel's signal — and every later copy
    // of the process on that engine, the fetcher's record copies first of all, waits behind it: the fetch stage read 2-3.6 ms per
    // 537 M samples for 4.6 MB of records per chunk (r04g-r04p), whatever the chain ran beside.
#if MGPU_EXPERIMENTS
    if (c->fsum_wide) {
        HIPCHK(c, hipMemsetAsync(r.d, 0, 2 * nb * sizeof(double), c->stream_f));
        launch_fsum_sc16_wide(cfg.format, iq, sl.d_mag, sl.n, cfg.buf_samples, r.d, r.d + nb, cfg.mode_ac ? 1 : 0, r.scratch, c->stream_f);
        HIPCHK(c, hipMemcpyAsync(r.h, r.d, 2 * nb * sizeof(double), hipMemcpyDeviceToHost, c->stream_f));
    } else
#endif
    launch_fsum_sc16(cfg.format, iq, sl.n, cfg.buf_samples, r.h, r.h + nb, cfg.mode_ac ? 1 : 0, c->stream_f, 1);
    HIPCHK(c, hipEventRecord(r.ev, c->stream_f));
    sl.fsum_pending = true;
    return MGPU_OK;
}

// The converter of this chunk on stream_c, beside the k_slice of the chunk before?  UC8 without Mode A/C (its scan wants the sums at
// once) and not a shard pass (their chunks come one at a time).
// Converter and sweep in one kernel for this chunk?  UC8 samples (the table's format), no Mode A/C (its scan wants the magnitudes and
// the sums before the sweep), not a struct mag_buf entry (the magnitudes are the caller's).
static bool sweep_is_fused(const mgpu_ctx *c, const Slot &sl) {
    const uint32_t buf_steps = c->cfg.buf_samples / (uint32_t) kSweepTile;      // (a power of two of steps per buffer: the kernel tests a step's place in its buffer with a mask)
    // (sweep_fused bit 0: UC8 through the shared table, k_sweep_uc8; bit 1: the SC16 formats by arithmetic, k_sweep_sc16)
    const int bit = c->cfg.format == MGPU_FMT_UC8 ? 1 : 2;
    return (c->sweep_fused & bit) && !c->cfg.mode_ac && !sl.have_mag && buf_steps && (buf_steps & (buf_steps - 1u)) == 0u;
}

static bool convert_on_side(const mgpu_ctx *c, const Slot &sl) {
    return !sweep_is_fused(c, sl) && c->conv_side && c->cfg.format == MGPU_FMT_UC8 && !c->cfg.mode_ac && !sl.have_mag && c->shard_mode == 0;
}

static int enqueue_convert(mgpu_ctx *c, Slot &sl, const uint8_t *iq) {
    const mgpu_config &cfg = c->cfg;
    const uint64_t n = sl.n;
    const bool side = convert_on_side(c, sl);
    hipStream_t s = side ? c->stream_c : c->stream;
    sl.timed = c->timing_every <= 1 || (c->timing_seq++ % (uint64_t) c->timing_every) == 0;
    sl.fsum_pending = false;
    // the slot's magnitudes / class bitmap / message lists are still read by the window-statistics
    // kernel of its previous use (stream2)
    if (sl.window_pending) { HIPCHK(c, hipStreamWaitEvent(s, sl.ev_window, 0)); if (side) HIPCHK(c, hipStreamWaitEvent(c->stream, sl.ev_window, 0)); sl.window_pending = false; }
    if (side) {
        // behind the k_sweep of the chunk before: everything the main stream ran with this slot's buffers (its chunk of kSlots ago) is
        // through by then, and so is the converter that wrote the 326-sample tail (this stream).  Then beside that chunk's k_slice.
        const Slot &prev = c->slot[(sl.seq + mgpu_ctx::kSlots - 1) % mgpu_ctx::kSlots];
        if (sl.seq > 0 && prev.swept_seq.load(std::memory_order_acquire) == sl.seq - 1) HIPCHK(c, hipStreamWaitEvent(s, prev.ev_swept, 0));
    }
    // (the scratch block is zero: k_publish of the slot's previous chunk left it so)
    sl.fused_iq = nullptr;
    if (sl.timed && !sweep_is_fused(c, sl)) HIPCHK(c, hipEventRecord(sl.ev[0], s));   // (fused: no converter to bracket; ev[1] opens k_sweep_uc8's)
    if (sweep_is_fused(c, sl)) {                            // nothing to launch: k_sweep_uc8 reads the samples itself
        sl.fused_iq = iq;
        sl.fused_tail = c->tail_src;
        sl.fsum_iq = iq;
        c->tail_src = n >= (uint64_t) kTrailing ? sl.d_mag + n : nullptr;
    } else if (!sl.have_mag) {
        ConvertParams cp{};
        cp.iq = iq; cp.mag = sl.d_mag; cp.n = n; cp.buf_samples = cfg.buf_samples;
        cp.tail = c->tail_src;          // the 326 magnitudes before this chunk (sdr_ifile.c:209-213), read in place
        cp.uc8_folded = c->d_uc8_folded;
        cp.sum_level = sl.d_sum_level; cp.sum_power = sl.d_sum_power;
        cp.fsum_level = sl.d_fsum_level; cp.fsum_power = sl.d_fsum_power;
        launch_convert(cfg.format, cp, s, side ? c->conv_side_blocks : 0u, c->convert_variant);
        if (cfg.format != MGPU_FMT_UC8 && cfg.mode_ac) {
            // mean level / power of the SC16 formats = the reference's sequential float sums (k_fsum_*, kernels/convert.inc), on a stream
            // of their own, into buffers of their own.  Mode A/C needs them before its scan (the noise floor): behind the converter
            // (they predict from its magnitudes), at once; the main stream waits for them below
            HIPCHK(c, hipEventRecord(sl.ev_pre, s));
            int rc = enqueue_fsum(c, sl, iq, sl.ev_pre);
            if (rc != MGPU_OK) return rc;
        }
        sl.fsum_iq = iq;
        // lastbuf->length < trailing_samples -> zeros (only possible for a stream shorter than 326 samples)
        c->tail_src = n >= (uint64_t) kTrailing ? sl.d_mag + n : nullptr;
    } else {
        c->tail_src = sl.d_mag + n;
    }
    if (cfg.mode_ac && sl.have_mag && sl.have_noise) {   // struct mag_buf entry: one buffer, noise level from the caller's means
        HIPCHK(c, hipMemcpyAsync(sl.d_ac_noise, &sl.given_noise, sizeof(uint32_t), hipMemcpyHostToDevice, s));
        launch_modeac_scan(sl.d_mag, n, cfg.buf_samples, sl.d_ac_noise, sl.h_ac, (uint32_t) c->cap_ac,
                           sl.d_scratch + CNT_NUM + 1 + 4 * c->cap_buffers, sl.d_counters, s);
    }
    if (cfg.mode_ac && !sl.have_mag && sl.fsum_pending) HIPCHK(c, hipStreamWaitEvent(s, c->fsum_ring[sl.fsum_idx].ev, 0));
    if (cfg.mode_ac && !sl.have_mag)       // Mode A/C candidates (needs the converter's per-buffer sums); a few us, streaming
        launch_modeac(sl.d_mag, n, cfg.buf_samples, cfg.format, sl.d_sum_level, sl.d_sum_power,
                      sl.fsum_pending ? c->fsum_ring[sl.fsum_idx].h : sl.d_fsum_level, sl.fsum_pending ? c->fsum_ring[sl.fsum_idx].h + c->cap_buffers : sl.d_fsum_power,
                      sl.d_ac_noise, sl.h_ac, (uint32_t) c->cap_ac, sl.d_scratch + CNT_NUM + 1 + 4 * c->cap_buffers, sl.d_counters, s);
    if (sl.timed) HIPCHK(c, hipEventRecord(sl.ev[1], s));
    if (side) {                                              // the main stream goes on when the magnitudes are there
        HIPCHK(c, hipEventRecord(sl.ev_conv, s));
        HIPCHK(c, hipStreamWaitEvent(c->stream, sl.ev_conv, 0));
        if (sl.timed) HIPCHK(c, hipEventRecord(sl.ev_sweep0, c->stream));
    }
    return MGPU_OK;
}

static int enqueue_sweep(mgpu_ctx *c, Slot &sl) {
    const mgpu_config &cfg = c->cfg;
    const uint64_t n = sl.n;
    const uint32_t nunits = (uint32_t) ((n + kUnit - 1) / kUnit);
    hipStream_t s = c->stream;
    SweepParams sp{};
    sp.mag = sl.d_mag; sp.n = n; sp.thr = sl.thr;
    sp.valid_long = c->valid_long; sp.valid_short = c->valid_short;
    sp.fix_df = (cfg.fixDF && cfg.nfix_crc) ? 1 : 0;
    sp.bit_syndrome = c->d_bit_syndrome; sp.parity = c->d_parity; sp.group_syndrome = c->d_group_syndrome;
    sp.tab_long = c->d_tab_long; sp.tab_short = c->d_tab_short; sp.n_long = c->n_long; sp.n_short = c->n_short;
    sp.pool = sl.d_pool; sp.pool_cap = (uint32_t) c->cap_pool; sp.pool_used = sl.d_pool_used;
    sp.unit_first = sl.d_unit_first; sp.unit_count = sl.d_unit_count; sp.nunits = nunits; sp.dealer = sl.d_dealer;
    sp.adder_bitmap = c->d_adder_bitmap; sp.counters = sl.d_counters;
    sp.cand = sl.d_cand; sp.cand_count = sl.d_cand_count; sp.sweep_part = sl.d_sweep_part;
    if (sl.fused_iq) {
        sp.iq = sl.fused_iq; sp.iq_format = (uint32_t) cfg.format; sp.tail = sl.fused_tail; sp.uc8_sym = c->d_uc8_folded + UC8_SYM_OFFSET; sp.mag_w = sl.d_mag;
        sp.sum_level = sl.d_sum_level; sp.sum_power = sl.d_sum_power;
        sp.buf_steps = cfg.buf_samples / (uint32_t) kSweepTile;
    }
    // ev[1] (recorded behind the converter, enqueue_convert) .. ev[4] bracket exactly one kernel: k_sweep (bench.py's roofline); ev[4] .. ev[2]: k_slice
#if MGPU_EXPERIMENTS
    sp.debug_stage = c->dbg_stage;     // (k_slice's leave-out experiments, tools/slice_stages.sh)
#endif
    {
        sl.sweep_blocks = launch_sweep(sp, s);
        if (sl.timed) HIPCHK(c, hipEventRecord(sl.ev[4], s));
        HIPCHK(c, hipEventRecord(sl.ev_swept, s));
        sl.swept_seq.store(sl.seq, std::memory_order_release);
        // Without Mode A/C nobody needs the float sums before the builder's statistics: their chain (one wave per buffer for ~0.35 ms at
        // s_setprio 3; beside k_sweep it doubled that kernel's time: 59 against 33 us per 512 buffers, round 3) starts when the chunk's
        // k_sweep is through and runs beside k_slice.  (Behind k_slice instead — beside the post-sweep kernels and the next chunk's
        // converter and k_sweep: 173-183 Gsamples/s against 191-203, gpurun r04q.)
        if (cfg.format != MGPU_FMT_UC8 && !cfg.mode_ac && !sl.have_mag && sl.fsum_iq) {
            const int rc = enqueue_fsum(c, sl, sl.fsum_iq, sl.ev_swept);
            if (rc != MGPU_OK) return rc;
        }
        sl.slice_blocks = launch_slice(sp, s, c->slice_blocks_cap);      // (0 = whatever is resident; experiments build: MGPU_SLICE_BLOCKS)
    }
    if (sl.timed) HIPCHK(c, hipEventRecord(sl.ev[2], s));
    return MGPU_OK;
}

static int enqueue_post(mgpu_ctx *c, Slot &sl) {
    const uint64_t n = sl.n;
    const uint32_t nunits = (uint32_t) ((n + kUnit - 1) / kUnit);
    hipStream_t s = c->stream;

    // class planes -> class bitmap, pre-screen (the surviving records stay in HBM: d_live), counters and per-buffer sums to the host
    PostSweepParams q{};
    q.pool = sl.d_pool; q.pool_cap = (uint32_t) c->cap_pool; q.variant = c->prescreen_variant; q.unit_first = sl.d_unit_first; q.first_count = sl.d_unit_count; q.nunits = nunits; q.chains_per_unit = (uint32_t) (kUnit / 2048); q.adder_bitmap = c->d_adder_bitmap;
    q.unit_live = sl.d_unit_live; q.block_live = sl.d_unit_live + c->cap_units + 2; q.live = sl.d_live; q.live_sig = sl.d_live_sig; q.counters = sl.d_counters;
    // a shard pass hands its records to another rank, which has no samples: their signal powers go with them.  Otherwise they are
    // computed after the walk, for the accepted frames only (k_msg_sig): 40 % of the work, off the main stream
    sl.sig_late = (c->sig_late && c->shard_mode == 0) || c->shard_mode == 3;   // (mode 3 wants the records only: clock estimates)
    q.mag = sl.sig_late ? nullptr : sl.d_mag;
    q.class_final = sl.d_class_final;
    q.class_words = (n + 31) / 32;
    q.dealer = sl.d_dealer;
    q.cand = sl.d_cand; q.cand_count = sl.d_cand_count;
    q.live_win = c->shard_mode == 2 ? sl.d_live_win : nullptr; q.n = n; q.thr = sl.thr; q.buf_len = c->cfg.buf_samples;
    q.d_scratch = sl.d_scratch; q.h_scratch = sl.h_scratch; q.scratch_words = (uint32_t) (sl.scratch_bytes / sizeof(unsigned long long));
    // the count pass leaves its decisions as masks in the segment headers (one scoring pass = one segment of at most 64
    // records), so the write pass does not look at the adder bitmap again
    q.fin_part = q.block_live + c->cap_units / 4 + 2;
    q.slice_part = sl.d_sweep_part; q.slice_blocks = sl.slice_blocks;       // k_slice's rows of counts (0 rows: the experiments build's fused kernel counts for itself)
    hipStream_t s_write = c->stream_pw ? c->stream_pw : s;
    if (c->stream_pw && c->post_beside == 2) {           // experiment: the whole post-sweep stage beside the next chunk's sweep
        HIPCHK(c, hipEventRecord(sl.ev_scan, s));
        HIPCHK(c, hipStreamWaitEvent(s_write, sl.ev_scan, 0));
        s = s_write;
    }
    if (launch_prescreen(q, s, s_write, sl.ev_scan) != 0) { c->err = "event ordering of the pre-screen passes failed"; return MGPU_E_HIP; }
    if (sl.timed) HIPCHK(c, hipEventRecord(sl.ev[3], s_write));
    HIPCHK(c, hipEventRecord(sl.ev_done, s_write));
    return MGPU_OK;
}

// one chunk on its own (struct mag_buf entry, shard passes)
static int enqueue_slot(mgpu_ctx *c, Slot &sl, const uint8_t *iq, hipEvent_t after_convert = nullptr) {
    int rc = enqueue_convert(c, sl, iq);
    // `after_convert`: the chunk's IQ samples have been read — by the converter AND, for the SC16 formats, by the float sums on the
    // second stream, which run beside the converter (Mode A/C) or start behind the chunk's k_sweep (enqueue_sweep)
    // (converter and sweep in one kernel: it is the sweep that reads the samples — the event behind enqueue_convert, which launches
    // nothing then, would let the next upload into this region of the staging buffer run beside it)
    const bool fsum_late = (c->cfg.format != MGPU_FMT_UC8 && !c->cfg.mode_ac && !sl.have_mag) || sweep_is_fused(c, sl);
    auto mark_read = [&]() -> int {
        if (sl.fsum_pending) {
            HIPCHK(c, hipEventRecord(sl.ev_convdone, c->stream));
            HIPCHK(c, hipStreamWaitEvent(c->stream_f, sl.ev_convdone, 0));
            HIPCHK(c, hipEventRecord(after_convert, c->stream_f));
        } else HIPCHK(c, hipEventRecord(after_convert, convert_on_side(c, sl) ? c->stream_c : c->stream));
        return MGPU_OK;
    };
    if (rc == MGPU_OK && after_convert && !fsum_late) rc = mark_read();
    if (rc == MGPU_OK) rc = enqueue_sweep(c, sl);
    if (rc == MGPU_OK && after_convert && fsum_late) rc = mark_read();
    if (rc == MGPU_OK) rc = enqueue_post(c, sl);
    return rc;
}

// ---- host half of a chunk, part 1 (fetcher thread): wait for the GPU, copy the live records out of pinned memory ----
// The chunk's live records and their would-be signal powers, HBM -> the job's ordinary memory.
static void unpin_records(mgpu_ctx *c, Slot &sl, HostJob &job, bool team) {
    const uint64_t nlive = job.nlive;
    job.recs.resize(nlive + 1);
    job.recs[nlive].pos = 0xFFFFFFFFu;          // sentinel for the walk
    if (!sl.sig_late) job.sig.resize(nlive);
    const int parts = team && nlive >= 65536 ? c->walk_threads : 1;
    auto copy = [&](int i) {
        const uint64_t lo = nlive * (uint64_t) i / parts, hi = nlive * (uint64_t) (i + 1) / parts;
        std::memcpy(job.recs.data() + lo, sl.h_live + lo, (hi - lo) * sizeof(PhaseRec));
        if (!sl.sig_late) std::memcpy(job.sig.data() + lo, sl.h_live_sig + lo, (hi - lo) * sizeof(unsigned long long));
    };
    if (parts > 1) c->walk_team.run(parts, copy); else copy(0);
    if (c->shard_mode == 2) job.win.assign(sl.h_live_win, sl.h_live_win + nlive);
    job.fetched = true;
}

static int fetch_records(mgpu_ctx *c, Slot &sl, HostJob &job, Slot *next) {
    const uint64_t nlive = job.nlive;
    // exactly nlive records + signal powers, HBM -> page-locked host memory over the copy engine (its own stream: the next
    // chunk's kernels keep running), then into ordinary memory: page-locked memory the device wrote is slow for the walk's
    // small scattered reads (4x slower walk) but streams at tens of GB/s (0.1 ms for 70 k records)
    // The copies run as blit kernels (the runtime's choice on this box, whatever HSA_ENABLE_SDMA / GPU_FORCE_BLIT_COPY_SIZE say)
    // and cost whatever is on the main stream meanwhile ~25 us: the converter takes 78 us instead of 52, or — held back until the
    // next chunk's converter and k_sweep are through (measured in round 2) — k_slice 140 instead of 114.  The same either way
    // (2.23 vs 2.27 ms per step), so: at once.
    if (nlive) {
        // Round 6: the chunk is complete the moment the NEXT chunk's sweep starts on the main stream, and since that sweep is
        // k_sweep_uc8 — no converter in front of it any more — the copy's blit kernel (83 us for 2.2 MB) lands on it every time.
        // Holding the copy behind that sweep (d2h_hold, experiments build: MGPU_D2H_HOLD=1) protects the sweep but puts ~0.1 ms per
        // chunk of waiting into the fetch stage (1.21 of a 1.23 ms feed); a hardware queue of its own for this stream (mgpu_create: the
        // side streams) steadies the sweep without that: off.
        if (c->d2h_hold && next && next->swept_seq.load(std::memory_order_acquire) == sl.seq + 1 && hipEventQuery(next->ev_swept) != hipSuccess)
            HIPCHK(c, hipStreamWaitEvent(c->stream_d2h, next->ev_swept, 0));
        // (a copy kernel of our own with 8..32 workgroups in place of the runtime's blit kernel: 2.55-2.63 ms per step instead of 2.45)
        HIPCHK(c, hipMemcpyAsync(sl.h_live, sl.d_live, nlive * sizeof(PhaseRec), hipMemcpyDeviceToHost, c->stream_d2h));
        if (!sl.sig_late) HIPCHK(c, hipMemcpyAsync(sl.h_live_sig, sl.d_live_sig, nlive * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream_d2h));
        if (c->shard_mode == 2) HIPCHK(c, hipMemcpyAsync(sl.h_live_win, sl.d_live_win, nlive * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream_d2h));
        HIPCHK(c, wait_stream_spin(c->stream_d2h));
    }
    if (!c->dump_dir.empty()) {   // replay material for tools/walk_replay.cpp
        const char *dd = c->dump_dir.c_str();
        static int dumped = 0;
        if (!dumped++) {
            std::string base = std::string(dd) + "/walk_";
            FILE *f = fopen((base + "recs.bin").c_str(), "wb"); fwrite(sl.h_live, sizeof(PhaseRec), nlive, f); fclose(f);
            f = fopen((base + "sig.bin").c_str(), "wb"); fwrite(sl.h_live_sig, 8, nlive, f); fclose(f);
            f = fopen((base + "bufs.bin").c_str(), "wb"); fwrite(sl.buffers.data(), sizeof(BufferClock), sl.buffers.size(), f); fclose(f);
        }
    }
    // Out of the page-locked buffer into ordinary memory: GPU-written pinned memory is 4x slower for the walk's small scattered reads.
    // (Round 4 tried the copy on the walker's team instead, in parallel: it takes as long there — reading the pinned pages is the
    // cost, not the copying thread — and the walker is the stage with less slack: dense bursts 2.1-2.7 ms of walk per 537 M samples
    // against 1.4, the rate unchanged.  So: here.)
    unpin_records(c, sl, job, false);
    return MGPU_OK;
}

static int fetch_slot(mgpu_ctx *c, Slot &sl, HostJob &job, Slot *next) {
    HIPCHK(c, wait_event_spin(sl.ev_done));
    const double t_gpu_done = wall_ms();
    if (sl.h_counters[CNT_POOL_OVERFLOW]) {
        c->err = "record pool overflow: recreate the context with a larger record_pool_records";
        return MGPU_E_OVERFLOW;
    }
    float ms;
    if (sl.timed) {
        if (!sl.fused_iq && hipEventElapsedTime(&ms, sl.ev[0], sl.ev[1]) == hipSuccess) c->acc.convert_ms += ms;
        if (hipEventElapsedTime(&ms, convert_on_side(c, sl) ? sl.ev_sweep0 : sl.ev[1], sl.ev[4]) == hipSuccess) { c->acc.sweep_ms += ms; sweep_pace_feedback(ms * 1e3f, sl.n, sl.sweep_blocks, c->event_bracket_us, !sl.fused_iq ? 0 : c->cfg.format == MGPU_FMT_UC8 ? 1 : 2); }
        if (hipEventElapsedTime(&ms, sl.ev[4], sl.ev[2]) == hipSuccess) c->acc.slice_ms += ms;
        if (hipEventElapsedTime(&ms, sl.ev[2], sl.ev[3]) == hipSuccess) c->acc.prescreen_ms += ms;
        c->acc.n_timed_chunks += 1;
        if (sl.fused_iq) c->acc.sweep_fused_chunks += 1.0f;
    }
    const uint64_t nlive = sl.h_counters[CNT_LIVE_TOTAL];

    const double t_f0 = wall_ms();
    if (nlive > c->cap_pool) { c->err = "live record count beyond the pool"; return MGPU_E_OVERFLOW; }
    job.nlive = nlive;
    job.stream_pos = sl.stream_pos;
    job.fetched = job.from_device = false;
    job.sig_late = sl.sig_late;
    // with the walk on the device the records stay in HBM (the walker fetches them itself for a chunk it has to walk here)
    // (... and a shard's first pass wants the adder bitmap only: its records are pre-screened against half a bitmap and go nowhere)
    if ((c->device_walk != 1 || c->shard_mode != 0) && c->shard_mode != 1) { const int rc = fetch_records(c, sl, job, next); if (rc != MGPU_OK) return rc; }
    job.ac.clear();
    if (c->cfg.mode_ac && (!sl.have_mag || sl.have_noise)) {
        const unsigned long long *counts = sl.h_scratch + CNT_NUM + 1 + 4 * c->cap_buffers;   // k_modeac's kAcLists lists
        const uint64_t cap_l = c->cap_ac / kAcLists;
        for (int l = 0; l < kAcLists; ++l) {
            if (counts[l] > cap_l) { c->err = "Mode A/C candidate buffer overflow"; return MGPU_E_OVERFLOW; }
            job.ac.insert(job.ac.end(), sl.h_ac + (size_t) l * cap_l, sl.h_ac + (size_t) l * cap_l + counts[l]);
        }
    }
    job.buffers = sl.buffers;
    job.given_mean_power = sl.given_mean_power;
    job.sums.assign(sl.h_sums, sl.h_sums + 2 * c->cap_buffers);
    // the float sums have their own pace: a chain of ~0.4 ms per chunk that starts behind the chunk's k_sweep.  Only the noise statistics
    // need them, so it is the builder that waits (build_job), two pipeline stages later — unless this is a shard pass, whose packets
    // carry them: then here
    job.fsum_idx = -1;
    if (sl.fsum_pending) {
        sl.fsum_pending = false;
        if (c->shard_mode == 0) job.fsum_idx = sl.fsum_idx;
        else {
            HIPCHK(c, wait_event_spin(c->fsum_ring[sl.fsum_idx].ev));
            std::memcpy(sl.h_fsums, c->fsum_ring[sl.fsum_idx].h, 2 * c->cap_buffers * sizeof(double));
        }
    }
    if (job.fsum_idx < 0) job.fsums.assign(sl.h_fsums, sl.h_fsums + 2 * c->cap_buffers);
    // ---- counters that do not depend on the skip windows ----
    const unsigned long long *hc = sl.h_counters;
    if (!(c->shard_stream && job.stream_pos < c->shard_stream_own_first)) {   // (a rank's warm-up is walked for the filter's state only: its statistics are nobody's)
        c->feed_cand[0] += hc[CNT_CANDIDATES];
        for (int i = 0; i < 5; ++i) c->feed_cand[1 + i] += hc[CNT_PHASE0 + i];
        c->feed_cand[6] += hc[CNT_CLASS_COND];
        c->feed_cand[7] += hc[CNT_CLASS_UNCOND];
    }
    c->acc.n_candidates += hc[CNT_CANDIDATES];
    c->acc.n_records += hc[CNT_RECORDS];
    c->acc.n_live_records += nlive;
    c->acc.n_chunks += 1;
    c->acc.d2h_ms += (float) (wall_ms() - t_f0);
    if (c->dbg_print) fprintf(stderr, "dbg: timeline: gpu done %.3f, fetched %.3f\n", t_gpu_done - c->feed_t0, wall_ms() - c->feed_t0);
    return MGPU_OK;
}

#if MGPU_EXPERIMENTS
// ---- the ordered walk on the device (kernels/walk.inc) ----
// Enqueues the walk of the slot's chunk on `s` against the filter as it stands now.  false = this chunk is not one the device
// walk models (a buffer longer than the context's buffer size, generations too large for the input blob): the host walks it.
static bool device_walk_enqueue(mgpu_ctx *c, Slot &sl, uint64_t nlive, hipStream_t s) {
    const uint32_t nbuf = (uint32_t) sl.buffers.size();
    const std::vector<uint32_t> &act = c->resolver.filter().members(true), &ina = c->resolver.filter().members(false);
    if (nbuf + 1 > c->cap_buffers + 1 || walk_input_bytes(nbuf, (uint32_t) act.size(), (uint32_t) ina.size()) > c->wk_in_cap) return false;
    if (nbuf == 0 || sl.buffers[0].length == 0 || sl.buffers[0].length > c->cfg.buf_samples) return false;
    const uint32_t L = sl.buffers[0].length;                  // the grid the kernels assume: buffer b = positions [b * L, b * L + length <= L)
    for (uint32_t b = 0; b < nbuf; ++b)
        if (sl.buffers[b].first != (uint64_t) b * L || sl.buffers[b].length > L || (b + 1 < nbuf && sl.buffers[b].length != L)) return false;
    WalkIn in{};
    in.next_flip = c->resolver.next_flip();
    in.nbuf = nbuf; in.n_active = (uint32_t) act.size(); in.n_inactive = (uint32_t) ina.size(); in.acc_cap = c->wk_acc_cap;
    in.nlive = (uint32_t) nlive;
    in.buf_len = L;
    in.pad[0] = c->wk_serial_only;
    uint8_t *p = c->h_wk_in;
    std::memcpy(p, &in, sizeof(in));
    std::memcpy(p + kWkInHead, sl.buffers.data(), (size_t) nbuf * sizeof(BufferClock));
    uint32_t *lists = (uint32_t *) (p + kWkInHead + (size_t) nbuf * sizeof(BufferClock));
    if (!act.empty()) std::memcpy(lists, act.data(), act.size() * sizeof(uint32_t));
    if (!ina.empty()) std::memcpy(lists + act.size(), ina.data(), ina.size() * sizeof(uint32_t));
    launch_device_walk(c->h_wk_in, sl.d_wk_in, walk_input_bytes(nbuf, in.n_active, in.n_inactive), c->wk, sl.d_live, nbuf,
                       c->h_wk_sum, sl.d_wk_acc, sl.d_msg_pos, sl.d_msg_limit, sl.d_msg_skip, (uint32_t) c->cap_msgs, s);
    return hipEventRecord(c->ev_wk, s) == hipSuccess;
}

static void counts_from_device(const WalkSummary &ws, ResolveCounts &rc) {
    const unsigned long long *k = ws.counts;   // WKC_* (kernels/walk.inc)
    rc.visited_groups = k[0]; rc.rejected_unknown = k[1]; rc.rejected_bad = k[2];
    for (int i = 0; i < 3; ++i) rc.accepted[i] = k[3 + i];
    for (int i = 0; i < 5; ++i) rc.best_phase[i] = k[6 + i];
    rc.skipped_uncond_groups = k[11]; rc.skipped_cond_groups = k[12]; rc.visited_cond_groups = k[13]; rc.visited_uncond_groups = k[14];
}

// check mode: the device walk of the chunk (enqueued before the host walk, against the same state) against what the host decided
static int device_walk_compare(mgpu_ctx *c, Slot &sl, HostJob &job, uint32_t nmsg) {
    HIPCHK(c, hipEventSynchronize(c->ev_wk));
    const WalkSummary &ws = *(const WalkSummary *) c->h_wk_sum;
    const uint32_t nbuf = (uint32_t) sl.buffers.size();
    const uint32_t *per_buf = (const uint32_t *) (c->h_wk_sum + sizeof(WalkSummary)), *adds = per_buf + 6 * (size_t) nbuf;
    c->wk_stats[5] += ws.iterations;
    if (ws.bad) for (int t = 0; t < 2; ++t) HIPCHK(c, hipMemsetAsync(c->wk.first[t], 0xff, sizeof(uint32_t) << 24, c->stream2));
    if (!ws.converged || ws.bad) { c->wk_stats[2] += 1; if (c->dbg_print) fprintf(stderr, "dbg: device walk: converged %u bad %u after %u walks\n", ws.converged, ws.bad, ws.iterations); return MGPU_OK; }
    const bool applied = c->wk_shadow.apply_device_walk(per_buf, adds, nbuf, ws.flip);
    if (!applied) { c->wk_stats[3] += 1; if (c->dbg_print) fprintf(stderr, "dbg: device walk: premises failed (table grew or the expiry moved)\n"); return MGPU_OK; }
    c->wk_stats[1] += 1;
    uint64_t bad = 0;
    if (ws.nmsg != nmsg) ++bad;
    else if (nmsg) {
        std::vector<Accepted> dev(nmsg);
        std::vector<uint32_t> pos(nmsg), limit(nmsg);
        std::vector<uint16_t> skip(nmsg);
        HIPCHK(c, hipMemcpy(dev.data(), sl.d_wk_acc, (size_t) nmsg * sizeof(Accepted), hipMemcpyDeviceToHost));
        HIPCHK(c, hipMemcpy(pos.data(), sl.d_msg_pos, (size_t) nmsg * 4, hipMemcpyDeviceToHost));
        HIPCHK(c, hipMemcpy(limit.data(), sl.d_msg_limit, (size_t) nmsg * 4, hipMemcpyDeviceToHost));
        HIPCHK(c, hipMemcpy(skip.data(), sl.d_msg_skip, (size_t) nmsg * 2, hipMemcpyDeviceToHost));
        for (uint32_t i = 0; i < nmsg; ++i) {
            const Accepted &h = job.acc[i], &d = dev[i];
            if (h.rec != d.rec || h.buffer != d.buffer || h.score != d.score || pos[i] != job.pos[i] || limit[i] != c->w_limit[i] || skip[i] != c->w_skip[i]) ++bad;
        }
    }
    ResolveCounts rc;
    counts_from_device(ws, rc);
    if (std::memcmp(&rc, &job.rc, sizeof(rc)) != 0) ++bad;
    if (!c->wk_shadow.same_state(c->resolver)) ++bad;
    c->wk_stats[6] += bad;
    if (bad || c->dbg_print) fprintf(stderr, "%s: device walk: %u walks, %u messages (host %u), %u adds, %llu differences\n", bad ? "mgpu" : "dbg", ws.iterations, ws.nmsg, nmsg, ws.nadds_total, (unsigned long long) bad);
    return MGPU_OK;
}

// MGPU_DEVICE_WALK=1: the chunk's walk on the device; the host checks its premises, catches the filter up, and has the messages
// and the window statistics made from the accept list where it lies.  MGPU_E_AGAIN = the host has to walk this chunk.
static constexpr int MGPU_E_AGAIN = -1000;
static int walk_job_device(mgpu_ctx *c, Slot &sl, HostJob &job) {
    const double t_res0 = wall_ms();
    hipStream_t s2 = c->stream2;                            // what follows the walk: behind it by ev_wk, but not in the next walk's way
    c->wk_stats[0] += 1;
    if (!device_walk_enqueue(c, sl, job.nlive, c->stream_wk)) { c->wk_stats[4] += 1; return MGPU_E_AGAIN; }
    const double t_enq = wall_ms();
    HIPCHK(c, hipEventSynchronize(c->ev_wk));
    const double t_gpu = wall_ms();
    const WalkSummary &ws = *(const WalkSummary *) c->h_wk_sum;
    const uint32_t nbuf = (uint32_t) sl.buffers.size();
    const uint32_t *per_buf = (const uint32_t *) (c->h_wk_sum + sizeof(WalkSummary)), *adds = per_buf + 6 * (size_t) nbuf;
    c->wk_stats[5] += ws.iterations;
    if (ws.bad) {     // a list overflowed: the tables may hold entries nothing remembers
        for (int t = 0; t < 2; ++t) HIPCHK(c, hipMemsetAsync(c->wk.first[t], 0xff, sizeof(uint32_t) << 24, c->stream_wk));
    }
    if (!ws.converged || ws.bad) { c->wk_stats[2] += 1; return MGPU_E_AGAIN; }
    const bool applied = c->resolver.apply_device_walk(per_buf, adds, nbuf, ws.flip);
    if (c->dbg_print) fprintf(stderr, "dbg: device walk: enqueue %.3f ms, gpu %.3f ms, apply %.3f ms (%d); %u walks, %u msgs, %u adds\n", t_enq - t_res0, t_gpu - t_enq, wall_ms() - t_gpu, (int) applied, ws.iterations, ws.nmsg, ws.nadds_total);
    if (!applied) { c->wk_stats[3] += 1; return MGPU_E_AGAIN; }
    c->wk_stats[1] += 1;
    const uint32_t nmsg = ws.nmsg;
    if (nmsg > c->cap_msgs) { c->err = "max_messages exceeded"; return MGPU_E_OVERFLOW; }
    job.rc = ResolveCounts();
    counts_from_device(ws, job.rc);
    c->feed_rc.add(job.rc);
    job.buf_nacc.resize(nbuf);
    for (uint32_t b = 0; b < nbuf; ++b) job.buf_nacc[b] = per_buf[6 * (size_t) b];
    c->acc.resolve_ms += (float) (wall_ms() - t_res0);

    const double t_sig0 = wall_ms();
    if (nmsg)
        launch_window_stats(sl.d_mag, sl.n, sl.d_cand, sl.d_cand_count, sl.d_class_final, sl.d_msg_pos, sl.d_msg_skip, sl.d_msg_limit, nmsg, sl.d_win_part, c->d_win, s2);
    const bool to_device_list = c->device_msgs == 1 && job.feed >= 0;   // (mode 2 with the walk on the device: not offered — mgpu_set_device_messages refuses it)
    if (nmsg) {
        const void *d_bufs = sl.d_wk_in + kWkInHead;           // the buffer clocks went over with the walk's input
        mgpu_msg *dst = c->d_wk_msgs;
        if (to_device_list) {
            FeedSlot &fs = c->feed[job.feed];
            if (fs.d_count + nmsg > fs.d_list_cap) { c->err = "device message list of the feed is full"; return MGPU_E_OVERFLOW; }
            dst = fs.d_list + fs.d_count;
            fs.d_count += nmsg;
        }
        launch_msg_sig(sl.d_mag, sl.d_msg_pos, sl.d_msg_skip, nmsg, sl.d_wk_sig, nullptr, s2);
        launch_build_messages(sl.d_live, nullptr, sl.d_wk_sig, sl.d_wk_acc, d_bufs, nmsg, dst, s2);
        if (to_device_list) HIPCHK(c, hipEventRecord(c->feed[job.feed].ev_built, s2));
        else HIPCHK(c, hipMemcpyAsync(job.h_msgs, c->d_wk_msgs, (size_t) nmsg * sizeof(mgpu_msg), hipMemcpyDeviceToHost, s2));
        HIPCHK(c, hipMemcpyAsync(job.h_msig, sl.d_wk_sig, (size_t) nmsg * sizeof(unsigned long long), hipMemcpyDeviceToHost, s2));
        HIPCHK(c, hipEventRecord(job.ev_copied, s2));
        HIPCHK(c, hipEventRecord(sl.ev_window, s2));           // the slot's device side is read on stream2 until here
        sl.window_pending = true;
    }
    c->acc.sigpower_ms += (float) (wall_ms() - t_sig0);
    job.nmsg = nmsg;
    job.from_device = true;
    c->acc.n_messages += nmsg;
    return MGPU_OK;
}

#endif  // MGPU_EXPERIMENTS

// The ordered walk of one chunk's live records on the host: buffer ranges walked in parallel against the filter as it stands
// now and committed in stream order (resolve.h: Resolver::parallel_walk) — exact, serial only where speculation fails — or the
// plain serial walk for small chunks.  Decisions into job.acc / job.pos / c->w_limit / c->w_skip, counts into job.rc.
static int64_t host_walk(mgpu_ctx *c, HostJob &job, const PhaseRec *recs, const std::vector<BufferClock> &buffers, uint64_t nlive, uint64_t aux_cap) {
    int64_t wn;
    const uint32_t nbuf_all = (uint32_t) buffers.size();
    const int K = c->walk_ranges > 0 ? c->walk_ranges : c->walk_threads;      // ranges per round (the team takes them by ticket: Team::run)
    if (K >= 2 && c->walk_threads >= 2 && nbuf_all >= (uint32_t) (4 * K) && nlive >= 4096) {
        // buffer ranges walked in parallel against the filter as it stands now, committed in stream order
        // (resolve.h: Resolver::parallel_walk); exact, and serial only where speculation fails.
        // K ranges per ROUND of about kWalkRound buffers, whatever the chunk's length: the further a range lies from the state it
        // speculates against, the more often its assumptions fail — a chunk of 2048 buffers walked as K ranges of 256 took 1.45-1.54 ms
        // per 537 M samples where chunks of 1024 take 1.0-1.1 (gpurun r06ah), while the GPU side wants the longer chunk (fewer
        // launches and tails: 1.11 against 1.21 ms of kernels).
        constexpr uint32_t kWalkRound = 1024;
        const uint32_t rounds = nbuf_all >= kWalkRound + kWalkRound / 2 ? (nbuf_all + kWalkRound / 2) / kWalkRound : 1u;
        std::vector<SegmentWalk> &segs = c->segs;
        if ((int) segs.size() != K) segs.resize(K);
        uint64_t total = 0, batches = 0;
        double t_walk = 0, t_gather = 0;
        wn = 0;
        for (uint32_t r = 0; r < rounds && wn >= 0; ++r) {
            const uint32_t r_lo = (uint32_t) ((uint64_t) nbuf_all * r / rounds), r_hi = (uint32_t) ((uint64_t) nbuf_all * (r + 1) / rounds);
            for (int k = 0; k < K; ++k) {
                segs[k].b_lo = r_lo + (uint32_t) ((uint64_t) (r_hi - r_lo) * k / K);
                segs[k].b_hi = r_lo + (uint32_t) ((uint64_t) (r_hi - r_lo) * (k + 1) / K);
                segs[k].rec_lo = segs[k].b_lo == 0 ? 0 : segment_first_record(recs, nlive, buffers[segs[k].b_lo].first);
            }
            const uint64_t round_end = r_hi >= nbuf_all ? nlive : segment_first_record(recs, nlive, buffers[r_hi].first);
            for (int k = 0; k < K; ++k) segs[k].rec_hi = k + 1 < K ? segs[k + 1].rec_lo : round_end;
            const double tp0 = wall_ms();
            c->resolver.parallel_walk(recs, nlive, buffers, segs,
                                      [&](int ntasks, const std::function<void(int)> &task) { c->walk_team.run(ntasks, task); }, &batches);
            const double tp2 = wall_ms();
            uint64_t round_total = 0;
            for (int k = 0; k < K; ++k) {
                round_total += segs[k].nacc;
                job.rc.add(segs[k].counts);
            }
            c->spec_segments += (uint64_t) K;
            if (total + round_total > aux_cap) wn = -1;
            else {
                if (job.acc.size() < total + round_total) job.acc.resize(total + round_total);
                // every range's decisions to their place in the chunk's arrays: by the team, a range each (one thread copying all of
                // them was 0.06-0.1 ms of the 0.5 ms a chunk of 2048 buffers spends in the walker stage, gpurun r06an)
                uint64_t offs[64];
                { uint64_t off = total; for (int k = 0; k < K && k < 64; ++k) { offs[k] = off; off += segs[k].nacc; } }
                c->walk_team.run(K, [&](int k) {
                    const uint64_t m = segs[k].nacc, off = offs[k];
                    std::memcpy(job.acc.data() + off, segs[k].acc.data(), m * sizeof(Accepted));
                    std::memcpy(job.pos.data() + off, segs[k].pos.data(), m * sizeof(uint32_t));
                    std::memcpy(c->w_limit.data() + off, segs[k].limit.data(), m * sizeof(uint32_t));
                    std::memcpy(c->w_skip.data() + off, segs[k].skip.data(), m * sizeof(uint16_t));
                });
                total += round_total;
                wn = (int64_t) total;
            }
            t_walk += tp2 - tp0; t_gather += wall_ms() - tp2;
        }
        c->spec_batches += batches;
        if (c->dbg_print) fprintf(stderr, "dbg: %u round(s) of %d ranges in %llu batches: %.3f ms, gather %.3f ms\n", rounds, K, (unsigned long long) batches, t_walk, t_gather);
    } else {
        wn = c->resolver.decide(recs, nlive, buffers, job.acc, job.pos.data(), c->w_skip.data(),
                                c->w_limit.data(), aux_cap, job.rc);
    }
    return wn;
}

// The second stream's work of a chunk (k_stage_in, k_msg_sig, k_window_stats, k_window_reduce, k_build_messages: ~125 us of small
// latency-bound kernels) starts when the chunk's walk is done — at a random point of the main stream's convert | sweep | slice | post
// cycle of a later chunk.  Beside k_slice or the converter it costs next to nothing; beside k_sweep, the kernel that streams at
// half the HBM peak, it cost that kernel 4-5 of its 32 us (profiles/r03_slice_stages.txt: 31.7-32.0 us with the second stream idle,
// 36.1 with it; round 4 A/B in the benchmark, profiles/r04_s2_behind_sweep.txt: k_sweep 39.9-42.5 -> 31.7-32.2 us, the step 1.94 ->
// 1.80 ms; the stream's priority made no difference either way).  So: if the main stream is about to run, or is running, a later chunk's k_sweep (that chunk's converter has the
// stream: the chunk before it is complete, its own sweep is not), the second stream waits for that sweep's event and runs beside
// the k_slice that follows; otherwise it starts at once.  Timing only: no result depends on where these kernels run.
static int hold_behind_sweep(mgpu_ctx *c, const Slot &sl, int slot_idx, hipStream_t s2) {
    const Slot *prev = &sl;                                  // (its chunk is complete: the walker has its records)
    for (int d = 1; d < mgpu_ctx::kSlots; ++d) {
        const Slot &o = c->slot[(slot_idx + d) % mgpu_ctx::kSlots];
        if (o.swept_seq.load(std::memory_order_acquire) != sl.seq + (uint64_t) d) break;   // not enqueued (yet)
        if (hipEventQuery(o.ev_swept) == hipSuccess) { prev = &o; continue; }              // that sweep is behind us
        if (prev == &sl || hipEventQuery(prev->ev_done) == hipSuccess) HIPCHK(c, hipStreamWaitEvent(s2, o.ev_swept, 0));
        break;
    }
    return MGPU_OK;
}

// ---- part 2 (walker thread): the ordered accept walk, then the window statistics of what it hid ----
static void shard_mark_now(mgpu_ctx *c);
static int walk_job(mgpu_ctx *c, Slot &sl, HostJob &job) {
    const uint64_t n = sl.n;
    const uint64_t nlive = job.nlive;
    // a rank's pass through the pipeline (mgpu_shard_stream_*): the range begins with this chunk — or this chunk is still warm-up,
    // walked for the filter's state only: no statistics, nothing behind the walk
    if (c->shard_stream && !c->shard_marked && job.stream_pos >= c->shard_stream_own_first) shard_mark_now(c);
    const bool warmup = c->shard_stream && job.stream_pos < c->shard_stream_own_first;
    const double t_res0 = wall_ms();
    const uint64_t aux_cap = nlive + 1 < c->cap_msgs ? nlive + 1 : c->cap_msgs;
    job.pos.resize(aux_cap); c->w_limit.resize(aux_cap); c->w_skip.resize(aux_cap);
    job.rc = ResolveCounts();
    int64_t wn;
    if (!job.fetched && c->device_walk != 1) unpin_records(c, sl, job, true);
#if MGPU_EXPERIMENTS
    bool wk_running = false;
    if (c->device_walk == 1) {
        const int rc = walk_job_device(c, sl, job);
        if (rc != MGPU_E_AGAIN) return rc;
        // not this chunk (see mgpu_debug_device_walk): its records come over after all and it is walked here
        const int frc = fetch_records(c, sl, job, nullptr);
        if (frc != MGPU_OK) return frc;
        if (!job.fetched) unpin_records(c, sl, job, true);
    }
    if (c->device_walk == 2) {
        c->wk_stats[0] += 1;
        c->wk_shadow.copy_state(c->resolver);
        wk_running = device_walk_enqueue(c, sl, nlive, c->stream2);
        if (!wk_running) c->wk_stats[4] += 1;
    }
#endif
    wn = host_walk(c, job, job.recs.data(), sl.buffers, nlive, aux_cap);
    if (wn > 0) {                                            // (page-locked: what the second stream's kernels read)
        const int parts = wn >= 32768 ? c->walk_threads : 1;
        auto copy = [&](int i) {
            const size_t lo = (size_t) wn * (size_t) i / parts, hi = (size_t) wn * (size_t) (i + 1) / parts;
            std::memcpy(sl.h_msg_pos + lo, job.pos.data() + lo, (hi - lo) * sizeof(uint32_t));
            std::memcpy(sl.h_msg_limit + lo, c->w_limit.data() + lo, (hi - lo) * sizeof(uint32_t));
            std::memcpy(sl.h_msg_skip + lo, c->w_skip.data() + lo, (hi - lo) * sizeof(uint16_t));
        };
        if (parts > 1) c->walk_team.run(parts, copy); else copy(0);
    }
    c->acc.resolve_ms += (float) (wall_ms() - t_res0);
    if (c->dbg_print) fprintf(stderr, "dbg: walk %.3f ms for %llu live records -> %lld msgs\n", wall_ms() - t_res0, (unsigned long long) nlive, (long long) wn);
    if (wn < 0) { c->err = "max_messages exceeded"; return MGPU_E_OVERFLOW; }
    const uint32_t nmsg = warmup ? 0u : (uint32_t) wn;      // (warm-up: the builder skips the job, nobody wants its windows or signal powers)
    if (!warmup) c->feed_rc.add(job.rc);
#if MGPU_EXPERIMENTS
    if (wk_running) { const int rc = device_walk_compare(c, sl, job, nmsg); if (rc != MGPU_OK) return rc; }
#endif

    // what the skip windows hid from the counters: asynchronous on the second stream, totals are
    // accumulated on the device and read once at the end of the feed
    const double t_sig0 = wall_ms();
    hipStream_t s2 = c->s_post;
    if (nmsg && c->s2_hold) { const int rc = hold_behind_sweep(c, sl, job.slot, s2); if (rc != MGPU_OK) return rc; }
    if (nmsg)
        launch_stage_in(sl.h_msg_pos, sl.h_msg_limit, sl.h_msg_skip, sl.d_msg_pos, sl.d_msg_limit, sl.d_msg_skip, nmsg, s2);
    // what the skip windows hid from the counters and — first, the builder waits for them — the accepted frames' signal powers, now
    // that it is known which frames they are: one kernel since round 5 (the window's samples are the frame's).  The kernel stores the
    // builder's copy itself: a hipMemcpyAsync from this thread contends with the fetcher's inside the runtime.
    if (nmsg) {
        launch_window_stats(sl.d_mag, n, sl.d_cand, sl.d_cand_count, sl.d_class_final, sl.d_msg_pos, sl.d_msg_skip,
                            sl.d_msg_limit, nmsg, sl.d_win_part, c->d_win, s2, job.sig_late ? sl.d_msg_sig : nullptr, job.sig_late ? job.h_msig : nullptr,
                            job.sig_late ? job.ev_copied : nullptr);
    }
    if (nmsg && c->device_msgs && job.feed >= 0) {
        // the accepted frames become message records on the device (kernels/build.inc), appended to the feed's device list
        FeedSlot &fs = c->feed[job.feed];
        const bool to_host = c->device_msgs == 2;
        if (fs.d_count + nmsg > fs.d_list_cap || (to_host && fs.d_count + nmsg > (uint64_t) fs.msgs.cap)) { c->err = to_host ? "the caller's message buffer (mgpu_set_message_buffer) is full" : "device message list of the feed is full"; return MGPU_E_OVERFLOW; }
        const size_t acc_bytes = ((size_t) nmsg * sizeof(Accepted) + 15) & ~(size_t) 15;
        const size_t buf_bytes = sl.buffers.size() * sizeof(BufferClock);
        std::memcpy(sl.h_blob, job.acc.data(), (size_t) nmsg * sizeof(Accepted));
        std::memcpy(sl.h_blob + acc_bytes, sl.buffers.data(), buf_bytes);
        launch_stage_blob(sl.h_blob, sl.d_blob, acc_bytes + buf_bytes, s2);
        launch_build_messages(sl.d_live, sl.d_live_sig, job.sig_late ? sl.d_msg_sig : nullptr, sl.d_blob, sl.d_blob + acc_bytes, nmsg,
                              fs.d_list + fs.d_count, s2);
        // mode 2: ... and on into the caller's page-locked array.  (k_build_messages storing there itself — 4.7 MB per chunk of 64-byte
        // stores over PCIe from a kernel on the second stream — stretched whatever ran beside it: 1.78 against 1.42 ms per feed, r06d.)
        if (to_host) HIPCHK(c, hipMemcpyAsync(fs.msgs.p + fs.d_count, fs.d_list + fs.d_count, (size_t) nmsg * sizeof(mgpu_msg), hipMemcpyDeviceToHost, s2));
        HIPCHK(c, hipEventRecord(fs.ev_built, s2));
        fs.d_count += nmsg;
    }
    if (nmsg) {          // the slot's device side is read on stream2 until here
        HIPCHK(c, hipEventRecord(sl.ev_window, s2));
        sl.window_pending = true;
    }
    c->acc.sigpower_ms += (float) (wall_ms() - t_sig0);

    job.nmsg = nmsg;
    if (c->dbg_print) fprintf(stderr, "dbg: timeline: walk %.3f .. %.3f\n", t_res0 - c->feed_t0, wall_ms() - c->feed_t0);
    c->acc.n_messages += nmsg;
    return MGPU_OK;
}

// ---- part 3 (builder thread): messages, signal and noise statistics ----------
static int build_job(mgpu_ctx *c, HostJob &job) {
    const mgpu_config &cfg = c->cfg;
    const double t0 = wall_ms();
    const uint32_t nmsg = job.nmsg;
    const uint32_t nbuf = (uint32_t) job.buffers.size();
    double stats_wait_ms = 0, wait_ms = 0;                         // idle: waiting for the GPU's second stream / the float sums' chain
    auto wait_copied = [&]() -> hipError_t { const double tw = wall_ms(); const hipError_t e = wait_event_spin(job.ev_copied); wait_ms += wall_ms() - tw; return e; };
    // Mode A/C (cfg.mode_ac): candidates in position order; an accepted reply hides the next 69 positions of its
    // buffer (f1_sample += 20 * 87 / 25, then the loop's ++, demod_2400.c:765)
    std::vector<AcCand> &ac = job.ac;
    std::vector<uint32_t> ac_buf;                            // buffer index of every accepted reply
    if (!ac.empty()) {
        std::sort(ac.begin(), ac.end(), [](const AcCand &x, const AcCand &y) { return x.pos < y.pos; });
        size_t keep = 0, i = 0;
        for (uint32_t b = 0; b < nbuf; ++b) {
            const uint64_t end = (uint64_t) job.buffers[b].first + job.buffers[b].length;
            uint64_t next_ok = 0;
            for (; i < ac.size() && ac[i].pos < end; ++i) {
                if (ac[i].pos < next_ok) continue;
                next_ok = (uint64_t) ac[i].pos + 20 * 87 / 25 + 1;
                ac[keep++] = ac[i];
                ac_buf.push_back(b);
            }
        }
        ac.resize(keep);
    }
    const uint32_t nac = (uint32_t) ac.size();
    if (c->shard_stream && job.stream_pos < c->shard_stream_own_first) return MGPU_OK;   // a rank's warm-up: walked for the filter's state only
    const bool on_device = c->device_msgs && job.feed >= 0;   // the walker had k_build_messages make the records: statistics only here
    MsgBuf &pending = job.feed >= 0 ? c->feed[job.feed].msgs : c->pending;
    const size_t first_msg = pending.size();
    if (!on_device && !pending.grow_for((size_t) nmsg + nac)) {
        c->err = pending.external ? "the caller's message buffer (mgpu_set_message_buffer) is full" : "out of memory for the decoded messages";
        return pending.external ? MGPU_E_OVERFLOW : MGPU_E_NOMEM;
    }
    const double t1 = wall_ms();
    mgpu_msg *out = pending.data() + first_msg;
    std::vector<mgpu_msg> &stage = c->b_stage;               // with Mode A/C the Mode S messages are built here and merged per buffer
    if (nac) { stage.resize(nmsg); }
    if (job.from_device) {                                   // the walk ran on the device: the messages come from there
        if (nmsg) HIPCHK(c, wait_copied());
        if (!on_device && nmsg) {
            mgpu_msg *dst = nac ? stage.data() : out;
            const int parts = nmsg >= 4096 ? c->build_threads : 1;
            c->build_team.run(parts, [&](int i) {
                const uint64_t lo = (uint64_t) nmsg * i / parts, hi = (uint64_t) nmsg * (i + 1) / parts;
                std::memcpy(dst + lo, job.h_msgs + lo, (hi - lo) * sizeof(mgpu_msg));
            });
        }
    } else {
        job.buf_nacc.assign(nbuf, 0);
        for (uint32_t i = 0; i < nmsg; ++i) job.buf_nacc[job.acc[i].buffer]++;
    }
    // per-message signal level, per-buffer noise power (demod_2400.c:436-457, 474-479): in stream order (double sums are order-dependent)
    auto build_statistics = [&]() {
        mgpu_counters &k = c->counters;
        const double *fsums = job.fsums.data();
        if (job.fsum_idx >= 0) {                                  // SC16 formats: the chunk's float sums arrive here at the latest
            const double tw = wall_ms();
            (void) wait_event_spin(c->fsum_ring[job.fsum_idx].ev);
            stats_wait_ms = wall_ms() - tw;
            fsums = c->fsum_ring[job.fsum_idx].h;
        }
        for (int i = 0; i < 3; ++i) k.demod_accepted[i] += job.rc.accepted[i];
        for (int i = 0; i < 5; ++i) k.demod_bestPhase[i] += job.rc.best_phase[i];
        // per-message signal level, per-buffer noise power (demod_2400.c:436-457, 474-479)
        uint32_t mi = 0;
        for (uint32_t b = 0; b < nbuf; ++b) {
            const BufferClock &bc = job.buffers[b];
            uint64_t sum_scaled = 0;
            for (const uint32_t mend = mi + job.buf_nacc[b]; mi < mend;) {
                unsigned long long sumsq;
                unsigned sig_len;                                 // msglen * 12 / 5, demod_2400.c:439
                if (job.from_device || job.sig_late) { sumsq = job.h_msig[mi] & ~(1ull << 63); sig_len = (job.h_msig[mi] >> 63) ? 268u : 134u; }
                else {
                    const uint32_t ri = job.acc[mi].rec;          // not from the message: those went out with streaming stores
                    sumsq = job.sig[ri];
                    sig_len = (job.recs[ri].msg[0] & 0x80) ? 268u : 134u;
                }
                const double signal_power = (double) sumsq / 65535.0 / 65535.0;
                const double level = signal_power / sig_len;
                k.signal_power_sum += signal_power;
                k.signal_power_count += sig_len;
                sum_scaled += sumsq;
                if (level > k.peak_signal_power) k.peak_signal_power = level;
                if (level > 0.50119) k.strong_signal_count++;
                ++mi;
            }
            double mean_power;
            if (!job.given_mean_power.empty()) mean_power = job.given_mean_power[b];
            else if (cfg.format == MGPU_FMT_UC8) mean_power = (double) job.sums[c->cap_buffers + b] / 65535.0 / 65535.0 / bc.length;   // convert.c:105-107
            else mean_power = (double) ((float) fsums[c->cap_buffers + b] / (float) bc.length);   // convert.c:246-248: a float sum, a float division
            const double sum_signal_power = (double) sum_scaled / 65535.0 / 65535.0;
            k.noise_power_sum += (mean_power * bc.length - sum_signal_power);
            if (c->shard_noise_on) c->shard_noise.push_back(mean_power * bc.length - sum_signal_power);
            k.noise_power_count += bc.length;
            k.samples_processed += bc.length;
            k.samples_lost += cfg.buf_samples - bc.length;        // readsb.c:886
            k.nbuffers++;
        }
        // a rank's pass: the messages' signal-power numerators, 8 bytes each, for the sum blocks (mgpu_shard_signal_terms) — appended in
        // bulk behind the chain above (inside it, one push_back per message cost the pass 1.8 of its 5.9 ms: profiles/r06_config5.txt)
        if (c->shard_noise_on && nmsg) {
            const size_t at = c->shard_sig.size();
            c->shard_sig.resize(at + nmsg);
            if (job.from_device || job.sig_late) for (uint32_t i = 0; i < nmsg; ++i) c->shard_sig[at + i] = job.h_msig[i] & ~(1ull << 63);
            else for (uint32_t i = 0; i < nmsg; ++i) c->shard_sig[at + i] = job.sig[job.acc[i].rec];
        }
    };
    bool stats_done = false;
    if (!on_device && !job.from_device) {
        const int parts = nmsg >= 4096 ? c->build_threads : 1;
        mgpu_msg *dst = nac ? stage.data() : out;
        // (sig_late: the accepted frames' signal powers come over the second stream — k_msg_sig and a copy, ahead of the window
        // statistics.  Building first and filling the field in afterwards was slower: the messages leave with streaming stores,
        // and touching 27 000 of their lines again costs more than the wait.)
        if (job.sig_late && nmsg) HIPCHK(c, wait_copied());
        // The signal / noise statistics — one chain of dependent double additions over the chunk's messages, a third of this stage's
        // time on one thread — ride beside the message build as one more task of the team (they read the accept list and the signal
        // powers only): with chunks of 1024 buffers the builder was the pipeline's slowest stage (round 4: 1.6-1.77 ms per step
        // against 1.59 ms of kernels).
        // (Round 6, last session, measured and NOT adopted — profiles/r06_builder.txt: per chunk of 140 k messages the chain takes 0.33 ms
        // on the calling thread, five ranges run beside it on the helpers for 0.2 ms each and the sixth behind them: 0.45 ms.  With the
        // statistics' divisions moved into the ranges, the chain reduced to the ordered additions and four ranges per thread taken by
        // ticket the stage takes 0.34 ms — build 0.76 instead of 0.98 ms per feed — and the FEED is 2 % longer in nine of ten interleaved
        // pairs, 1.13-1.15 against 1.113 ms: the GPU's post-sweep bracket grows from 0.232 to 0.248 ms per feed, the fetcher's stage with
        // it.  Six threads bursting through the messages with streaming stores for a third of a millisecond are worse neighbours to the
        // fetcher on the same L3 group, and to k_publish's writes into page-locked memory, than five threads and a chain taking their time.)
        const bool split = parts > 1;
        double task_ms[8][2] = {};                                 // (MGPU_DEBUG_PRINT: the tasks' begin and end)
        c->build_team.run(parts + (split ? 1 : 0), [&](int i) {
            const double tb = c->dbg_print ? wall_ms() : 0.0;
            if (split && i == 0) build_statistics();
            else {
                const int p = split ? i - 1 : i;
                const uint64_t lo = (uint64_t) nmsg * p / parts, hi = (uint64_t) nmsg * (p + 1) / parts;
                Resolver::build_messages(job.recs.data(), job.sig_late ? nullptr : job.sig.data(), job.sig_late ? job.h_msig + lo : nullptr, job.buffers,
                                         job.acc.data() + lo, hi - lo, dst + lo);
            }
            if (c->dbg_print && i < 8) { task_ms[i][0] = tb - t1; task_ms[i][1] = wall_ms() - t1; }
        });
        if (c->dbg_print) {
            fprintf(stderr, "dbg: build tasks (begin-end ms; task 0 = the statistics):");
            for (int i = 0; i < parts + (split ? 1 : 0) && i < 8; ++i) fprintf(stderr, " %.3f-%.3f", task_ms[i][0], task_ms[i][1]);
            fprintf(stderr, "\n");
        }
        stats_done = split;
    } else if (!job.from_device && job.sig_late && nmsg) HIPCHK(c, wait_copied());   // (messages on the device: statistics only)
    if (nac) {   // netUseMessage order: per buffer the Mode S messages of demodulate2400, then the replies of demodulate2400AC
        size_t si = 0, ai = 0, o = 0;
        for (uint32_t b = 0; b < nbuf; ++b) {
            const BufferClock &bc = job.buffers[b];
            for (uint32_t k = 0; k < job.buf_nacc[b]; ++k) out[o++] = stage[si++];
            for (; ai < nac && ac_buf[ai] == b; ++ai) {
                mgpu_msg m;
                std::memset(&m, 0, sizeof(m));
                m.timestamp = bc.sampleTimestamp + ac[ai].f2_clock / 5;                               // :755, 60 MHz -> 12 MHz, at F2
                m.sysTimestamp = bc.sysTimestamp + (m.timestamp - bc.sampleTimestamp) / 12000;        // :758
                m.sig_len = 1;                                                                        // signalLevel stays 0
                m.msgtype = 77;                                                                       // DFTYPE_MODEAC, decodeModeAMessage (mode_ac.c:165-200)
                m.msgbits = 16;
                m.msg[0] = m.raw[0] = (uint8_t) (ac[ai].modeac >> 8);
                m.msg[1] = m.raw[1] = (uint8_t) ac[ai].modeac;
                m.addr = ac[ai].modeac & 0xFF7Fu;                                                     // low 24 bits of (ModeA & 0xFF7F) | MODES_NON_ICAO_ADDRESS
                out[o++] = m;
            }
        }
        c->counters.demod_modeac += nac;
    }
    if (!on_device) pending.n = first_msg + nmsg + nac;
    const double t2 = wall_ms();

    if (!stats_done) build_statistics();
    c->acc.build_ms += (float) (wall_ms() - t0 - wait_ms - stats_wait_ms);
    c->acc.build_wait_ms += (float) (wait_ms + stats_wait_ms);
    if (c->dbg_print) fprintf(stderr, "dbg: timeline: build %.3f .. %.3f\n", t0 - c->feed_t0, wall_ms() - c->feed_t0);
    if (c->dbg_print) fprintf(stderr, "dbg: build: grow %.3f ms, messages %.3f ms, statistics %.3f ms for %u msgs\n", t1 - t0, t2 - t1, wall_ms() - t2, nmsg);
    return MGPU_OK;
}

// Start / end of one API call that runs chunks: reset and then apply the skip-window corrections of
// the demod counters (see DESIGN.md §1 "Statistics without a candidate log").
static int feed_begin(mgpu_ctx *c) {
    if (c->accounting_open) return MGPU_OK;  // deferred feeds: the accounting opened by an earlier feed is still running
    { std::lock_guard<std::mutex> lk(c->mu); c->hot.store(true, std::memory_order_relaxed); }
    c->cv.notify_all();                      // the stage threads switch from sleeping to polling
    c->accounting_open = true;
    c->timing_seq = 0;                       // the first chunk of an accounting period carries the stage events: mgpu_timing has kernel figures however short the period
    c->acct_t0 = wall_ms();
    std::memset(&c->acc, 0, sizeof(c->acc));
    std::memset(c->feed_cand, 0, sizeof(c->feed_cand));
    c->feed_rc = ResolveCounts();
    HIPCHK(c, hipMemsetAsync(c->d_win, 0, 8 * sizeof(unsigned long long), c->s_post));
    return MGPU_OK;
}

static int feed_end(mgpu_ctx *c) {
    c->accounting_open = false;
    if (c->shard_mode != 0) return MGPU_OK;   // a shard pass produces no messages and no statistics here
    HIPCHK(c, hipMemcpyAsync(c->h_win, c->d_win, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->s_post));
    HIPCHK(c, wait_stream_spin(c->s_post));
    mgpu_counters &k = c->counters;
    k.nflips = c->resolver.nflips();
    const unsigned long long *hw = c->h_win;
    const ResolveCounts &rc = c->feed_rc;
    const uint64_t C = c->feed_cand[0], U = c->feed_cand[6], R = c->feed_cand[7];
    const uint64_t cW = hw[0], uW = hw[4];
    k.demod_preambles += C - cW;
    k.demod_preamblePhase[0] += c->feed_cand[1] - hw[1];
    k.demod_preamblePhase[1] += c->feed_cand[2] - hw[1];
    k.demod_preamblePhase[2] += c->feed_cand[3] - hw[2];
    k.demod_preamblePhase[3] += c->feed_cand[4] - hw[2];
    k.demod_preamblePhase[4] += c->feed_cand[5] - hw[3];
    // candidates without any record score -2 for sure; those hidden inside skip windows are not counted
    k.demod_rejected_bad += (C - U - R) - (cW - uW - rc.skipped_uncond_groups) + rc.rejected_bad;
    // conditional-only candidates: dead ones (address can never be known) + the visited live ones the walk rejected
    k.demod_rejected_unknown_icao += rc.rejected_unknown + (U - rc.visited_cond_groups - uW);
    return MGPU_OK;
}

// A C++ exception (std::bad_alloc from a vector growing with the traffic) must not leave a stage thread — that would
// terminate the process — nor cross the extern "C" boundary: it becomes MGPU_E_NOMEM for the feed that hit it.
static int guarded(mgpu_ctx *c, const std::function<int()> &f) {
    try {
        return f();
    } catch (const std::bad_alloc &) {
        c->err = "out of host memory";
        return MGPU_E_NOMEM;
    } catch (const std::exception &e) {
        c->err = std::string("internal error: ") + e.what();
        return MGPU_E_NOMEM;
    }
}

static void fetcher_main(mgpu_ctx *c) {
    (void) hipSetDevice(c->cfg.device);
    for (;;) {
        int idx, jidx, next_idx = -1;
        {
            std::unique_lock<std::mutex> lk(c->mu);
            stage_wait(c, lk, [&] { return c->stop || !c->queue.empty(); });
            if (c->queue.empty()) return;   // stop requested and nothing left
            idx = c->queue.front();
            next_idx = c->queue.size() >= 2 ? c->queue[1] : -1;
            jidx = (int) (c->job_seq++ % mgpu_ctx::kJobs);
            stage_wait(c, lk, [&] { return !c->job[jidx].busy; });
            c->job[jidx].busy = true;
        }
        Slot &sl = c->slot[idx];
        HostJob &job = c->job[jidx];
        job.slot = idx;
        job.feed = sl.feed;
        int rc = c->worker_rc == MGPU_OK ? guarded(c, [&] { return fetch_slot(c, sl, job, next_idx >= 0 ? &c->slot[next_idx] : nullptr); }) : c->worker_rc;   // after an error just drain
        if (rc == MGPU_OK && c->shard_mode == 3) {               // the pre-pass of the stream form: what the buffers' end clocks will be, nothing else
            c->shard_est_pos.push_back(job.stream_pos);
            c->shard_est_off.push_back(c->shard_est.size());
            estimate_end_clocks(job.recs.data(), job.nlive, sl.buffers, c->shard_est);
        }
        if (rc == MGPU_OK && c->shard_mode == 2) {
            // The chunk becomes a packet for the rank that walks: header, live records, per record its would-be signal power and
            // the counts of its would-be skip window, per buffer the converter's level / power sums — everything the statistics
            // of an unsharded run take from the samples, so that the walking rank needs none (kPacketWords: the header)
            const unsigned long long *hc = sl.h_counters;
            const uint64_t nbuf = sl.buffers.size();
            const uint64_t hdr[kPacketWords] = {job.stream_pos, sl.n, job.nlive, kPacketMagic, hc[CNT_CANDIDATES], hc[CNT_PHASE0 + 0], hc[CNT_PHASE0 + 2],
                                                hc[CNT_PHASE0 + 4], hc[CNT_CLASS_COND], hc[CNT_CLASS_UNCOND], nbuf, 0};
            std::vector<uint8_t> &pk = c->shard_packets;
            auto put = [&](const void *p8, size_t bytes) { pk.insert(pk.end(), (const uint8_t *) p8, (const uint8_t *) p8 + bytes); };
            c->shard_est_pos.push_back(job.stream_pos);
            c->shard_est_off.push_back(c->shard_est.size());
            // two things, side by side (the walker's team has nothing else to do during a shard pass): the packet, and what the buffers'
            // end clocks will be (mgpu_shard_clock_estimate) — on this thread alone they made the fetcher the pass's slowest stage
            c->walk_team.run(2, [&](int task) {
                if (task == 1) { estimate_end_clocks(job.recs.data(), job.nlive, sl.buffers, c->shard_est); return; }
                put(hdr, sizeof(hdr));
                put(job.recs.data(), (job.nlive + 1) * sizeof(PhaseRec));      // (with the walk's sentinel record)
                put(job.sig.data(), job.nlive * sizeof(unsigned long long));
                put(job.win.data(), job.nlive * sizeof(unsigned long long));
                // (UC8: exact integer sums; SC16*: the converter's double sums — eight bytes per buffer either way, level then power)
                if (c->cfg.format == MGPU_FMT_UC8) { put(sl.h_sums, nbuf * 8); put(sl.h_sums + c->cap_buffers, nbuf * 8); }
                else { put(sl.h_fsums, nbuf * 8); put(sl.h_fsums + c->cap_buffers, nbuf * 8); }
            });
        }
        {
            std::lock_guard<std::mutex> lk(c->mu);
            if (rc != MGPU_OK && c->worker_rc == MGPU_OK) c->worker_rc = rc;
            c->queue.pop_front();            // popped only now: wait_all sees the chunk at every stage
            if (rc == MGPU_OK && c->shard_mode == 0) c->walk_queue.push_back(jidx);
            else { job.busy = false; sl.busy = false; }       // error, or a shard pass: nothing to walk here
        }
        c->cv.notify_all();
    }
}

static void worker_main(mgpu_ctx *c) {
    (void) hipSetDevice(c->cfg.device);
    for (;;) {
        int jidx;
        {
            std::unique_lock<std::mutex> lk(c->mu);
            stage_wait(c, lk, [&] { return c->stop || !c->walk_queue.empty(); });
            if (c->walk_queue.empty()) return;
            jidx = c->walk_queue.front();
        }
        HostJob &job = c->job[jidx];
        Slot &sl = c->slot[job.slot];
        int rc = c->worker_rc == MGPU_OK ? guarded(c, [&] { return walk_job(c, sl, job); }) : c->worker_rc;
        {
            std::lock_guard<std::mutex> lk(c->mu);
            if (rc != MGPU_OK && c->worker_rc == MGPU_OK) c->worker_rc = rc;
            c->walk_queue.pop_front();
            sl.busy = false;                 // the GPU may have the slot back
            if (rc == MGPU_OK) c->build_queue.push_back(jidx); else job.busy = false;
        }
        c->cv.notify_all();
    }
}

static void builder_main(mgpu_ctx *c) {
    for (;;) {
        int jidx;
        {
            std::unique_lock<std::mutex> lk(c->mu);
            stage_wait(c, lk, [&] { return c->stop || !c->build_queue.empty(); });
            if (c->build_queue.empty()) return;
            jidx = c->build_queue.front();
        }
        int rc = guarded(c, [&] { return build_job(c, c->job[jidx]); });
        {
            std::lock_guard<std::mutex> lk(c->mu);
            if (rc != MGPU_OK && c->worker_rc == MGPU_OK) c->worker_rc = rc;
            c->build_queue.pop_front();      // popped only now: wait_all sees the job until it is done
            if (c->job[jidx].feed >= 0) c->feed[c->job[jidx].feed].jobs_built++;
            c->job[jidx].busy = false;
        }
        c->cv.notify_all();
    }
}

static Slot &acquire_slot(mgpu_ctx *c, int idx) {
    // The feeding thread waits the way the stage threads do (stage_wait: polling while the pipeline runs).  Asleep on the condition
    // variable it came back up to a millisecond after the walker had released the slot on the pool's loaded hosts — as long as the
    // three chunks the GPU still has queued take — and a run now and then settled at 2.25 ms per step with every kernel and every
    // stage as fast as ever (238 against 380 Gsamples/s, gpurun r05g): the pipeline's second slow steady state, after the stage
    // threads' sleeping GPU waits.
    std::unique_lock<std::mutex> lk(c->mu);
    stage_wait(c, lk, [&] { return !c->slot[idx].busy; });
    c->slot[idx].busy = true;
    return c->slot[idx];
}

static void submit_slot(mgpu_ctx *c, int idx) {
    { std::lock_guard<std::mutex> lk(c->mu); c->queue.push_back(idx); }
    c->cv.notify_all();
}

static int wait_all(mgpu_ctx *c) {
    std::unique_lock<std::mutex> lk(c->mu);
    stage_wait(c, lk, [&] { bool idle = true; for (const Slot &sl : c->slot) idle = idle && !sl.busy; return idle && c->queue.empty() && c->walk_queue.empty() && c->build_queue.empty(); });
    return c->worker_rc;
}

// buffer grid of ifileRun for `n` samples continuing at stream position `pos0` (sdr_ifile.c:194-241)
static void ifile_grid(const mgpu_ctx *c, uint64_t pos0, uint64_t n, std::vector<BufferClock> &v) {
    v.clear();
    const uint32_t B = c->cfg.buf_samples;
    for (uint64_t off = 0; off < n; off += B) {
        BufferClock b;
        const uint64_t len = n - off < B ? n - off : B;
        b.first = (uint32_t) off;
        b.length = (uint32_t) len;
        b.sampleTimestamp = (int64_t) (pos0 + off) * 5;                               // :206 (12 MHz / 2.4 MHz)
        b.sysTimestamp = b.sampleTimestamp / 12000 + c->cfg.startup_time_ms;         // :216
        v.push_back(b);
    }
}

static int feed_common(mgpu_ctx *c, const void *src, bool src_is_device, uint64_t n) {
    if (!c || (!src && n)) return MGPU_E_INVAL;
    if (n == 0) return MGPU_OK;
    if (c->eof) return MGPU_E_EOF;
    if (n > c->cap_samples) return MGPU_E_CAPACITY;
    if (c->worker_rc != MGPU_OK) return c->worker_rc;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    const size_t bps = c->cfg.format == MGPU_FMT_UC8 ? 2 : 4;
    const double t_start = wall_ms();
    if (!c->accounting_open) c->feed_t0 = t_start;
    int fidx = -1;
    if (c->deferred) {
        std::lock_guard<std::mutex> lk(c->mu);
        if (c->feed_tail - c->feed_head >= (uint64_t) mgpu_ctx::kFeeds) {
            c->err = "deferred feeds: collect the oldest feed before starting another (at most 4 uncollected)";
            return MGPU_E_INVAL;
        }
        fidx = (int) (c->feed_tail % mgpu_ctx::kFeeds);
        FeedSlot &fs = c->feed[fidx];
        fs.jobs_total = fs.jobs_built = 0;
        fs.closed = false;
        fs.msgs.clear();
        fs.d_count = 0;
        if (c->device_msgs == 2) {
            // the records go from the GPU straight into the array mgpu_set_message_buffer named for this feed: it has to be page-locked
            // (mgpu_host_alloc / mgpu_host_register), the kernel writes through its device address
            void *dp = nullptr;
            if (!fs.msgs.external || hipHostGetDevicePointer(&dp, fs.msgs.p, 0) != hipSuccess || !dp) {
                (void) hipGetLastError();            // (the runtime keeps the failed query as the thread's "last error": not ours to leave behind)
                c->err = "mgpu_set_device_messages(2): name a page-locked array (mgpu_host_alloc / mgpu_host_register) with mgpu_set_message_buffer before every feed";
                return MGPU_E_INVAL;
            }
            fs.host_dev = (mgpu_msg *) dp;
            if (!fs.ev_built && hipEventCreateWithFlags(&fs.ev_built, hipEventDisableTiming) != hipSuccess) { c->err = "device messages: event"; return MGPU_E_HIP; }
        }
        const bool ext_list = c->device_msgs == 1 && fs.d_ext;
        if (ext_list) { fs.d_list = fs.d_ext; fs.d_list_cap = fs.d_ext_cap; }
        fs.d_ext = nullptr; fs.d_ext_cap = 0;                // (named per feed)
        if (c->device_msgs && !fs.d_msgs && !ext_list) {
            const uint64_t want = ((c->cap_samples + c->chunk_samples - 1) / c->chunk_samples) * c->cap_msgs;   // every chunk of a feed may fill its slot's list
            if (hipSetDevice(c->cfg.device) != hipSuccess || hipMalloc(&fs.d_msgs, want * sizeof(mgpu_msg)) != hipSuccess ||
                hipEventCreateWithFlags(&fs.ev_built, hipEventDisableTiming) != hipSuccess) {
                c->err = "device message list: allocation failed";
                return MGPU_E_NOMEM;
            }
            fs.d_cap = want;
        }
        if (!ext_list) { fs.d_list = fs.d_msgs; fs.d_list_cap = fs.d_cap; }
        c->feed_tail++;                      // open: mgpu_collect sees it, and waits for `closed`
    }
    { int brc = feed_begin(c); if (brc != MGPU_OK) { c->hot.store(false, std::memory_order_relaxed); return brc; } }
    // host samples go up chunk by chunk on the copy stream, each chunk's convert waits for its own piece only:
    // the transfer of chunk i+1 overlaps the kernels of chunk i (fully so from pinned / mgpu_host_register'ed memory)
    const uint8_t *iq = src_is_device ? (const uint8_t *) src : c->d_iq;
    // software pipeline over chunks: GPU works on chunk i+1 while the worker walks chunk i
    int rc = MGPU_OK;
    hipEvent_t last_h2d = nullptr;
    for (uint64_t off = 0; off < n && rc == MGPU_OK; off += c->chunk_samples) {
        const uint64_t len = n - off < c->chunk_samples ? n - off : c->chunk_samples;
        const uint64_t seq = c->chunk_seq++;
        const int k = (int) (seq % mgpu_ctx::kSlots);
        Slot &sl = acquire_slot(c, k);
        sl.seq = seq;
        sl.n = len;
        sl.stream_pos = c->stream_pos + off;
        sl.feed = fidx;
        sl.have_mag = false;
        sl.have_noise = false;
        sl.thr = c->cfg.preamble_threshold;
        sl.given_mean_power.clear();
        ifile_grid(c, c->stream_pos + off, len, sl.buffers);
        hipEvent_t ev_read = nullptr;
        if (!src_is_device) {
            const double t0 = wall_ms();
            const size_t region = (size_t) (off / c->chunk_samples);
            hipError_t e = hipSuccess;
            // the previous chunk uploaded into this region of the staging buffer (an earlier feed) must have been converted
            if (c->iq_region_used[region]) e = hipStreamWaitEvent(c->stream_w, c->ev_iq_read[region], 0);
            ev_read = c->ev_iq_read[region];
            c->iq_region_used[region] = 1;
            if (e == hipSuccess) e = hipMemcpyAsync(c->d_iq + off * bps, (const uint8_t *) src + off * bps, len * bps, hipMemcpyHostToDevice, c->stream_w);
            if (e == hipSuccess) e = hipEventRecord(sl.ev_h2d, c->stream_w);
            if (e == hipSuccess) e = hipStreamWaitEvent(c->stream, sl.ev_h2d, 0);
            if (e == hipSuccess && c->conv_side) e = hipStreamWaitEvent(c->stream_c, sl.ev_h2d, 0);
            if (e != hipSuccess) { c->err = std::string("H2D of the IQ samples: ") + hipGetErrorString(e); rc = MGPU_E_HIP; }
            c->acc.h2d_ms += (float) (wall_ms() - t0);   // host time spent issuing (pageable memory: staging) the copies
        }
        if (fidx >= 0) { std::lock_guard<std::mutex> lk(c->mu); c->feed[fidx].jobs_total++; }
        if (rc == MGPU_OK) rc = enqueue_slot(c, sl, iq + off * bps, ev_read);
        if (rc != MGPU_OK) { std::lock_guard<std::mutex> lk(c->mu); if (c->worker_rc == MGPU_OK) c->worker_rc = rc; }
        last_h2d = src_is_device ? nullptr : sl.ev_h2d;
        submit_slot(c, k);   // even after an enqueue error: the stages release the slot
    }
    if (fidx >= 0) {
        // deferred: the feed is on its way; its messages are collected by mgpu_collect, the counters settle at the next drain
        { std::lock_guard<std::mutex> lk(c->mu); c->feed[fidx].closed = true; }
        c->cv.notify_all();
        // a deferred feed returns with its kernels still to run, but not with the caller's samples still to be read: the last
        // upload has landed when this returns (from page-locked memory the copies are truly asynchronous), so the caller may
        // reuse its block at once, as after a synchronous feed
        if (rc == MGPU_OK && last_h2d && wait_event_spin(last_h2d) != hipSuccess) { c->err = "H2D of the IQ samples failed"; rc = MGPU_E_HIP; }
        if (rc != MGPU_OK) return rc;
        c->stream_pos += n;
        if (n % c->cfg.buf_samples) c->eof = true;
        return MGPU_OK;
    }
    const int wrc = wait_all(c);
    c->hot.store(false, std::memory_order_relaxed);
    if (rc == MGPU_OK) rc = wrc;
    if (rc == MGPU_OK) rc = feed_end(c);
    if (rc != MGPU_OK) return rc;
    c->stream_pos += n;
    if (n % c->cfg.buf_samples) c->eof = true;   // short read = end of file (sdr_ifile.c:223-237)
    c->acc.total_ms = (float) (wall_ms() - t_start);
    c->timing = c->acc;
    return MGPU_OK;
}

// Deferred feeds: wait until nothing is in flight, then settle the counters (feed_end) and the timing of everything since
// the last drain.  Every call that reads or changes stream state other than feeding and collecting starts with this.
static int drain(mgpu_ctx *c) {
    if (!c->accounting_open) return c->worker_rc;
    (void) hipSetDevice(c->cfg.device);
    int rc = wait_all(c);
    c->hot.store(false, std::memory_order_relaxed);
    if (rc == MGPU_OK) rc = feed_end(c); else c->accounting_open = false;
    c->acc.total_ms = (float) (wall_ms() - c->acct_t0);
    c->timing = c->acc;
    return rc;
}

int mgpu_set_deferred(mgpu_ctx *c, int on) {
    if (!c) return MGPU_E_INVAL;
    const int rc = drain(c);
    if (rc != MGPU_OK) return rc;
    if (c->feed_head != c->feed_tail || c->pending.size()) { c->err = "mgpu_set_deferred: collect the pending messages first"; return MGPU_E_INVAL; }
    if (!on) {                                       // the caller's arrays go back to the caller
        for (auto &f : c->feed) { f.msgs.use_external(nullptr, 0); f.d_ext = nullptr; f.d_ext_cap = 0; }
        c->device_msgs = 0;
    }
    c->deferred = on != 0;
    return MGPU_OK;
}

int mgpu_feed_iq(mgpu_ctx *c, const void *iq_host, uint64_t nsamples) { return feed_common(c, iq_host, false, nsamples); }

int mgpu_selftest_device_index(const char *pci_devices_dir, const char *bus_id) {
    if (!pci_devices_dir || !bus_id) return -2;
    std::string base(pci_devices_dir);
    if (base.empty() || base.back() != '/') base += '/';
    return device_index_on_node(bus_id, base);
}

int mgpu_host_cpus(mgpu_ctx *c, int32_t *cpus, int32_t cap) {
    if (!c || (!cpus && cap)) return MGPU_E_INVAL;
    const int n = (int) c->host_cpus.size();
    for (int i = 0; i < n && i < cap; ++i) cpus[i] = c->host_cpus[i];
    return n;
}

void *mgpu_host_alloc(mgpu_ctx *c, uint64_t bytes) {
    if (!c || !bytes || hipSetDevice(c->cfg.device) != hipSuccess) return nullptr;
    NearDevice near(c->cfg.device);
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes) != hipSuccess) return nullptr;
    std::memset(p, 0, bytes);                    // first touch here, on the device's node
    return p;
}

void mgpu_host_free(mgpu_ctx *c, void *ptr) {
    if (c && ptr) { (void) hipSetDevice(c->cfg.device); (void) hipHostFree(ptr); }
}

int mgpu_host_register(mgpu_ctx *c, void *ptr, uint64_t bytes) {
    if (!c || !ptr || !bytes) return MGPU_E_INVAL;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    HIPCHK(c, hipHostRegister(ptr, bytes, hipHostRegisterDefault));
    return MGPU_OK;
}

int mgpu_host_unregister(mgpu_ctx *c, void *ptr) {
    if (!c || !ptr) return MGPU_E_INVAL;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    HIPCHK(c, hipHostUnregister(ptr));
    return MGPU_OK;
}

int mgpu_feed_iq_device(mgpu_ctx *c, const void *d_iq, uint64_t nsamples) { return feed_common(c, d_iq, true, nsamples); }

void *mgpu_device_iq_buffer(mgpu_ctx *c) { return c ? c->d_iq : nullptr; }

int mgpu_upload_iq(mgpu_ctx *c, const void *iq_host, uint64_t nsamples) {
    if (!c || !iq_host) return MGPU_E_INVAL;
    if (nsamples > c->cap_samples) return MGPU_E_CAPACITY;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    const size_t bps = c->cfg.format == MGPU_FMT_UC8 ? 2 : 4;
    HIPCHK(c, hipMemcpy(c->d_iq, iq_host, nsamples * bps, hipMemcpyHostToDevice));
    return MGPU_OK;
}

int mgpu_finish(mgpu_ctx *c) {
    if (!c) return MGPU_E_INVAL;
    { const int rc = drain(c); if (rc != MGPU_OK) return rc; }
    if (c->eof) return MGPU_OK;
    // file length an exact multiple of the buffer size: one more zero-length buffer, whose
    // converter call divides 0 by 0 (convert.c:101-107) -> noise_power_sum becomes NaN
    const int64_t st = (int64_t) c->stream_pos * 5;
    c->resolver.tick_empty(st / 12000 + c->cfg.startup_time_ms, st);
    c->counters.noise_power_sum += std::numeric_limits<double>::quiet_NaN();
    c->counters.samples_lost += c->cfg.buf_samples;
    c->counters.nbuffers++;
    c->counters.nflips = c->resolver.nflips();
    c->eof = true;
    return MGPU_OK;
}

static void take_messages(MsgBuf &mb, struct mgpu_msg *out, uint64_t cap, uint64_t *n) {
    const uint64_t k = mb.size() < cap ? mb.size() : cap;
    if (mb.external && out == mb.data() && k == mb.size()) {
        mb.clear();                          // the messages already are where the caller wants them
    } else {
        if (k) std::memcpy(out, mb.data(), k * sizeof(mgpu_msg));
        mb.drop_front(k);
    }
    if (n) *n = k;
}

int mgpu_collect(mgpu_ctx *c, struct mgpu_msg *out, uint64_t cap, uint64_t *n, struct mgpu_counters *counters) {
    if (!c || (!out && cap)) return MGPU_E_INVAL;
    if (n) *n = 0;
    if (c->deferred) {
        if (c->feed_head != c->feed_tail) {                  // the oldest uncollected feed: wait for its last chunk's messages only
            FeedSlot &fs = c->feed[c->feed_head % mgpu_ctx::kFeeds];
            {
                std::unique_lock<std::mutex> lk(c->mu);
                stage_wait(c, lk, [&] { return c->worker_rc != MGPU_OK || (fs.closed && fs.jobs_built == fs.jobs_total); });
                if (c->worker_rc != MGPU_OK) return c->worker_rc;
            }
            if (c->device_msgs == 2) {                       // built on the GPU, stored by it into the feed's (page-locked) array
                if (fs.d_count > cap) { c->err = "mgpu_collect: the feed's messages do not fit (device-messages mode takes whole feeds)"; return MGPU_E_OVERFLOW; }
                if (fs.d_count) {
                    HIPCHK(c, hipSetDevice(c->cfg.device));
                    HIPCHK(c, wait_event_spin(fs.ev_built));
                    if (out != fs.msgs.p) std::memcpy(out, fs.msgs.p, fs.d_count * sizeof(mgpu_msg));
                }
                if (n) *n = fs.d_count;
                std::lock_guard<std::mutex> lk(c->mu);
                c->feed_head++;
            } else if (c->device_msgs) {                     // the feed's records are in HBM: this entry copies them out
                if (fs.d_count > cap) { c->err = "mgpu_collect: the feed's messages do not fit (device-messages mode takes whole feeds)"; return MGPU_E_OVERFLOW; }
                if (fs.d_count) {
                    HIPCHK(c, hipSetDevice(c->cfg.device));
                    HIPCHK(c, wait_event_spin(fs.ev_built));
                    HIPCHK(c, hipMemcpy(out, fs.d_list, fs.d_count * sizeof(mgpu_msg), hipMemcpyDeviceToHost));
                }
                if (n) *n = fs.d_count;
                std::lock_guard<std::mutex> lk(c->mu);
                c->feed_head++;
            } else {
                take_messages(fs.msgs, out, cap, n);
                if (fs.msgs.size() == 0) { std::lock_guard<std::mutex> lk(c->mu); c->feed_head++; }
            }
        }
        if (counters) {                                      // exact counters need everything in flight to land: this drains
            const int rc = drain(c);
            if (rc != MGPU_OK) return rc;
            *counters = c->counters;
        }
        return MGPU_OK;
    }
    take_messages(c->pending, out, cap, n);
    if (counters) *counters = c->counters;
    return MGPU_OK;
}

int mgpu_collect_device(mgpu_ctx *c, const struct mgpu_msg **d_msgs, uint64_t *n, struct mgpu_counters *counters) {
    if (!c || !d_msgs || !n) return MGPU_E_INVAL;
    *d_msgs = nullptr; *n = 0;
    if (!c->deferred || c->device_msgs != 1) { c->err = "mgpu_collect_device: needs mgpu_set_deferred and mgpu_set_device_messages(1)"; return MGPU_E_INVAL; }
    if (c->feed_head != c->feed_tail) {
        FeedSlot &fs = c->feed[c->feed_head % mgpu_ctx::kFeeds];
        {
            std::unique_lock<std::mutex> lk(c->mu);
            stage_wait(c, lk, [&] { return c->worker_rc != MGPU_OK || (fs.closed && fs.jobs_built == fs.jobs_total); });
            if (c->worker_rc != MGPU_OK) return c->worker_rc;
        }
        if (fs.d_count) {
            HIPCHK(c, hipSetDevice(c->cfg.device));
            HIPCHK(c, wait_event_spin(fs.ev_built));
        }
        *d_msgs = fs.d_list; *n = fs.d_count;
        std::lock_guard<std::mutex> lk(c->mu);
        c->feed_head++;
    }
    if (counters) {
        const int rc = drain(c);
        if (rc != MGPU_OK) return rc;
        *counters = c->counters;
    }
    return MGPU_OK;
}

int mgpu_set_device_messages(mgpu_ctx *c, int on) {
    if (!c) return MGPU_E_INVAL;
    const int rc = drain(c);
    if (rc != MGPU_OK) return rc;
    if (c->feed_head != c->feed_tail) { c->err = "mgpu_set_device_messages: collect the pending feeds first"; return MGPU_E_INVAL; }
    if (on && (!c->deferred || c->cfg.mode_ac)) { c->err = "mgpu_set_device_messages: needs deferred feeds, and no Mode A/C (its replies are merged on the host)"; return MGPU_E_INVAL; }
    if (on < 0 || on > 2 || (on == 2 && c->device_walk == 1)) return MGPU_E_INVAL;
    if (on) {                                        // the four feeds' device lists now, not inside somebody's timed region
        HIPCHK(c, hipSetDevice(c->cfg.device));
        const uint64_t want = ((c->cap_samples + c->chunk_samples - 1) / c->chunk_samples) * c->cap_msgs;   // every chunk of a feed may fill its slot's list
        for (auto &fs : c->feed) {
            if (fs.d_msgs) continue;
            HIPCHK(c, hipMalloc(&fs.d_msgs, want * sizeof(mgpu_msg)));
            HIPCHK(c, hipEventCreateWithFlags(&fs.ev_built, hipEventDisableTiming));
            fs.d_cap = want;
        }
    }
    c->device_msgs = on;
    return MGPU_OK;
}

int mgpu_set_device_message_buffer(mgpu_ctx *c, struct mgpu_msg *d_buf, uint64_t capacity) {
    if (!c || (d_buf && !capacity)) return MGPU_E_INVAL;
    if (!c->deferred || c->device_msgs != 1) { c->err = "mgpu_set_device_message_buffer: needs mgpu_set_deferred and mgpu_set_device_messages(1)"; return MGPU_E_INVAL; }
    std::lock_guard<std::mutex> lk(c->mu);
    if (c->feed_tail - c->feed_head >= (uint64_t) mgpu_ctx::kFeeds) { c->err = "mgpu_set_device_message_buffer: collect the oldest feed first"; return MGPU_E_INVAL; }
    FeedSlot &fs = c->feed[c->feed_tail % mgpu_ctx::kFeeds];
    fs.d_ext = d_buf; fs.d_ext_cap = d_buf ? capacity : 0;
    return MGPU_OK;
}

int mgpu_set_message_buffer(mgpu_ctx *c, struct mgpu_msg *buf, uint64_t capacity) {
    if (!c || (buf && !capacity)) return MGPU_E_INVAL;
    if (c->deferred) {                                       // the array of the NEXT feed; earlier feeds keep theirs
        std::lock_guard<std::mutex> lk(c->mu);
        if (c->feed_tail - c->feed_head >= (uint64_t) mgpu_ctx::kFeeds) { c->err = "mgpu_set_message_buffer: collect the oldest feed first"; return MGPU_E_INVAL; }
        c->feed[c->feed_tail % mgpu_ctx::kFeeds].msgs.use_external(buf, (size_t) capacity);
        return MGPU_OK;
    }
    if (c->pending.size()) { c->err = "mgpu_set_message_buffer: collect the pending messages first"; return MGPU_E_INVAL; }
    c->pending.use_external(buf, (size_t) capacity);
    return MGPU_OK;
}

uint64_t mgpu_pending_messages(mgpu_ctx *c) {
    if (!c) return 0;
    if (!c->deferred) return c->pending.size();
    std::lock_guard<std::mutex> lk(c->mu);                   // deferred: messages of the feeds that are complete
    uint64_t total = 0;
    for (uint64_t f = c->feed_head; f != c->feed_tail; ++f) {
        const FeedSlot &fs = c->feed[f % mgpu_ctx::kFeeds];
        if (!(fs.closed && fs.jobs_built == fs.jobs_total)) break;
        total += c->device_msgs ? fs.d_count : fs.msgs.n;
    }
    return total;
}

int mgpu_filter_expire(mgpu_ctx *c) {
    if (!c) return MGPU_E_INVAL;
    if (c->cfg.filter_clock != MGPU_FILTER_CLOCK_EXTERNAL) {
        c->err = "mgpu_filter_expire: the context runs its own filter clock (cfg.filter_clock)";
        return MGPU_E_INVAL;
    }
    { const int rc = drain(c); if (rc != MGPU_OK) return rc; }
    c->resolver.external_expire();
    c->counters.nflips = c->resolver.nflips();
    return MGPU_OK;
}

int mgpu_filter_add(mgpu_ctx *c, uint32_t addr) {
    if (!c) return MGPU_E_INVAL;
    { const int rc = drain(c); if (rc != MGPU_OK) return rc; }
    c->resolver.filter().add(addr);
    return MGPU_OK;
}

int mgpu_debug_device_walk(mgpu_ctx *c, uint64_t out[8]) {
    if (!c || !out) return MGPU_E_INVAL;
    (void) drain(c);
    for (int i = 0; i < 8; ++i) out[i] = c->wk_stats[i];
    return MGPU_OK;
}

int mgpu_debug_last_magnitudes(mgpu_ctx *c, uint16_t *out, uint64_t n) {
    if (!c || (!out && n)) return MGPU_E_INVAL;
    { const int rc = drain(c); if (rc != MGPU_OK) return rc; }
    if (c->chunk_seq == 0) { c->err = "mgpu_debug_last_magnitudes: nothing fed yet"; return MGPU_E_INVAL; }
    const Slot &sl = c->slot[(c->chunk_seq - 1) % mgpu_ctx::kSlots];
    if (n > sl.n) { c->err = "mgpu_debug_last_magnitudes: more than the last chunk holds"; return MGPU_E_INVAL; }
    HIPCHK(c, hipSetDevice(c->cfg.device));
    HIPCHK(c, hipDeviceSynchronize());
    if (n) HIPCHK(c, hipMemcpy(out, sl.d_mag + kTrailing, n * sizeof(uint16_t), hipMemcpyDeviceToHost));
    return MGPU_OK;
}

int mgpu_event_bracket_us(mgpu_ctx *c, float *us) {
    if (!c || !us) return MGPU_E_INVAL;
    { const int rc = drain(c); if (rc != MGPU_OK) return rc; }
    HIPCHK(c, hipSetDevice(c->cfg.device));
    hipEvent_t a = nullptr, b = nullptr;
    HIPCHK(c, hipEventCreate(&a));
    if (hipEventCreate(&b) != hipSuccess) { (void) hipEventDestroy(a); c->err = "hipEventCreate failed"; return MGPU_E_HIP; }
    std::vector<float> v;
    // one workgroup per CU: every one of them is resident from the start, so the kernel lasts exactly what each of them spins
    // (a grid that does not fit spins in waves and reads longer than `known_us`: the bracket would come out inflated)
    hipDeviceProp_t prop;
    const unsigned blocks = hipGetDeviceProperties(&prop, c->cfg.device) == hipSuccess && prop.multiProcessorCount > 0 ? (unsigned) prop.multiProcessorCount : 256u;
    const unsigned known_us = 20;
    for (int r = 0; r < 24; ++r) {                  // | 50 us of something | ev | 20 us, exactly | ev | 50 us of something |
        launch_spin(50, blocks, nullptr, c->stream);
        (void) hipEventRecord(a, c->stream);
        launch_spin(known_us, blocks, nullptr, c->stream);
        (void) hipEventRecord(b, c->stream);
        launch_spin(50, blocks, nullptr, c->stream);
        if (hipStreamSynchronize(c->stream) != hipSuccess) break;
        float ms = 0;
        if (r >= 4 && hipEventElapsedTime(&ms, a, b) == hipSuccess) v.push_back(ms * 1e3f - (float) known_us);
    }
    (void) hipEventDestroy(a);
    (void) hipEventDestroy(b);
    if (v.empty()) { c->err = "mgpu_event_bracket_us: no measurement"; return MGPU_E_HIP; }
    std::sort(v.begin(), v.end());
    *us = v[v.size() / 2];
    if (*us > 0.5f && *us < 20.0f) c->event_bracket_us = *us;   // (k_sweep's pacing feedback takes it off the timed launches)
    return MGPU_OK;
}

int mgpu_last_timing(mgpu_ctx *c, struct mgpu_timing *t) {
    if (!c || !t) return MGPU_E_INVAL;
    (void) drain(c);
    *t = c->timing;
    return MGPU_OK;
}

int mgpu_convert(mgpu_ctx *c, const void *iq_host, uint16_t *mag_host, uint32_t n, double *out_mean_level, double *out_mean_power) {
    if (!c || !iq_host || !mag_host) return MGPU_E_INVAL;
    if (n > c->cap_samples || n > 0x7fffffffu) return MGPU_E_CAPACITY;
    { const int drc = drain(c); if (drc != MGPU_OK) return drc; }
    if (c->stream_pos != 0 && !c->eof) {
        // the converter borrows slot 0's magnitude buffer, which may hold the 326-sample tail the stream's next feed starts from
        c->err = "mgpu_convert: the context is in the middle of a stream (one context per role, or mgpu_reset first)";
        return MGPU_E_INVAL;
    }
    HIPCHK(c, hipSetDevice(c->cfg.device));
    const size_t bps = c->cfg.format == MGPU_FMT_UC8 ? 2 : 4;
    hipStream_t s = c->stream;
    Slot &sl = c->slot[0];
    double ml = std::numeric_limits<double>::quiet_NaN(), mp = ml;   // 0/0 for n == 0, as the reference
    // chunk by chunk through slot 0's magnitude buffer; one accumulation bucket for the whole call
    HIPCHK(c, hipMemsetAsync(sl.d_sum_level, 0, sizeof(unsigned long long), s));
    HIPCHK(c, hipMemsetAsync(sl.d_sum_power, 0, sizeof(unsigned long long), s));
    HIPCHK(c, hipMemsetAsync(sl.d_fsum_level, 0, sizeof(double), s));
    HIPCHK(c, hipMemsetAsync(sl.d_fsum_power, 0, sizeof(double), s));
    for (uint64_t off = 0; off < n; off += c->chunk_samples) {
        const uint64_t len = n - off < c->chunk_samples ? n - off : c->chunk_samples;
        HIPCHK(c, hipMemcpyAsync(c->d_iq, (const uint8_t *) iq_host + off * bps, len * bps, hipMemcpyHostToDevice, s));
        ConvertParams cp{};
        cp.iq = c->d_iq; cp.mag = sl.d_mag; cp.n = len;
        cp.buf_samples = 0x80000000u;
        cp.uc8_folded = c->d_uc8_folded;
        cp.sum_level = sl.d_sum_level; cp.sum_power = sl.d_sum_power;
        cp.fsum_level = sl.d_fsum_level; cp.fsum_power = sl.d_fsum_power;
        launch_convert(c->cfg.format, cp, s);
        launch_fsum_sc16(c->cfg.format, c->d_iq, len, 0x80000000u, sl.d_fsum_level, sl.d_fsum_power, 1, s);   // (the call's one bucket: the state carries over)
        HIPCHK(c, hipMemcpyAsync(mag_host + off, sl.d_mag + kTrailing, len * sizeof(uint16_t), hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipStreamSynchronize(s));
    }
    if (n) {
        HIPCHK(c, hipMemcpyAsync(sl.h_sums, sl.d_sum_level, sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipMemcpyAsync(sl.h_sums + 1, sl.d_sum_power, sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipMemcpyAsync(sl.h_fsums, sl.d_fsum_level, sizeof(double), hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipMemcpyAsync(sl.h_fsums + 1, sl.d_fsum_power, sizeof(double), hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipStreamSynchronize(s));
        if (c->cfg.format == MGPU_FMT_UC8) {
            ml = (double) sl.h_sums[0] / 65536.0 / n;              // convert.c:101-103 (sic, 65536)
            mp = (double) sl.h_sums[1] / 65535.0 / 65535.0 / n;    // convert.c:105-107
        } else {
            ml = (double) ((float) sl.h_fsums[0] / (float) n);      // convert.c:242-248: float sums, float divisions
            mp = (double) ((float) sl.h_fsums[1] / (float) n);
        }
    }
    // leave the slot's accumulation buckets zero again (the chunk pipeline relies on it)
    HIPCHK(c, hipMemsetAsync(sl.d_sum_level, 0, sizeof(unsigned long long), s));
    HIPCHK(c, hipMemsetAsync(sl.d_sum_power, 0, sizeof(unsigned long long), s));
    HIPCHK(c, hipMemsetAsync(sl.d_fsum_level, 0, sizeof(double), s));
    HIPCHK(c, hipMemsetAsync(sl.d_fsum_power, 0, sizeof(double), s));
    HIPCHK(c, hipStreamSynchronize(s));
    if (out_mean_level) *out_mean_level = ml;
    if (out_mean_power) *out_mean_power = mp;
    return MGPU_OK;
}

static int demod_mag_buf(mgpu_ctx *c, const uint16_t *data, uint32_t length, int64_t sampleTimestamp, int64_t sysTimestamp,
                         const double *mean_level, double mean_power, uint32_t dropped);

int mgpu_demod_mag_buf(mgpu_ctx *c, const uint16_t *data, uint32_t length, int64_t sampleTimestamp, int64_t sysTimestamp,
                       double mean_power, uint32_t dropped) {
    if (c && c->cfg.mode_ac) { c->err = "mode_ac needs the buffer's mean_level: use mgpu_demod_mag_buf_ac"; return MGPU_E_INVAL; }
    return demod_mag_buf(c, data, length, sampleTimestamp, sysTimestamp, nullptr, mean_power, dropped);
}

int mgpu_demod_mag_buf_ac(mgpu_ctx *c, const uint16_t *data, uint32_t length, int64_t sampleTimestamp, int64_t sysTimestamp,
                          double mean_level, double mean_power, uint32_t dropped) {
    return demod_mag_buf(c, data, length, sampleTimestamp, sysTimestamp, &mean_level, mean_power, dropped);
}

static int demod_mag_buf(mgpu_ctx *c, const uint16_t *data, uint32_t length, int64_t sampleTimestamp, int64_t sysTimestamp,
                         const double *mean_level, double mean_power, uint32_t dropped) {
    if (!c || !data) return MGPU_E_INVAL;
    if (length > c->chunk_samples) return MGPU_E_CAPACITY;
    if (c->deferred) { c->err = "deferred feeds are for the IQ entries (mgpu_feed_iq*): mgpu_set_deferred(ctx, 0) first"; return MGPU_E_INVAL; }
    if (c->worker_rc != MGPU_OK) return c->worker_rc;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    if (length == 0) {
        c->resolver.tick_empty(sysTimestamp);
        c->counters.noise_power_sum += mean_power * 0.0;
        c->counters.samples_lost += c->cfg.buf_samples;
        c->counters.nbuffers++;
        c->counters.nflips = c->resolver.nflips();
        return MGPU_OK;
    }
    const double t_start = wall_ms();
    { int brc = feed_begin(c); if (brc != MGPU_OK) return brc; }
    const uint64_t seq = c->chunk_seq++;
    const int slot_idx = (int) (seq % mgpu_ctx::kSlots);
    Slot &sl = acquire_slot(c, slot_idx);
    sl.seq = seq;
    sl.n = length;
    sl.feed = -1;
    // demod_2400.c:335-338: after dropped samples the reference raises the threshold to at least PREAMBLE_THRESHOLD_PIZERO
    sl.thr = dropped && c->cfg.preamble_threshold < 75 ? 75 : c->cfg.preamble_threshold;
    sl.have_mag = true;
    sl.have_noise = c->cfg.mode_ac && mean_level;
    if (sl.have_noise) {
        // demodulate2400AC, demod_2400.c:579-580: noise_stddev = sqrt(mean_power - mean_level^2); noise_level = (power + stddev) * 65535 + 0.5
        const double sd = std::sqrt(mean_power - *mean_level * *mean_level);
        sl.given_noise = (uint32_t) ((mean_power + sd) * 65535 + 0.5);
    }
    sl.given_mean_power.assign(1, mean_power);
    sl.buffers.assign(1, BufferClock{sampleTimestamp, sysTimestamp, 0u, length});
    int rc = MGPU_OK;
    {
        hipError_t e = hipMemcpyAsync(sl.d_mag, data, ((size_t) length + kTrailing) * sizeof(uint16_t), hipMemcpyHostToDevice, c->stream);
        if (e != hipSuccess) { c->err = std::string("H2D of the magnitude buffer: ") + hipGetErrorString(e); rc = MGPU_E_HIP; }
    }
    if (rc == MGPU_OK) rc = enqueue_slot(c, sl, nullptr);
    submit_slot(c, slot_idx);
    if (rc != MGPU_OK) { std::lock_guard<std::mutex> lk(c->mu); if (c->worker_rc == MGPU_OK) c->worker_rc = rc; }
    const int wrc = wait_all(c);
    c->hot.store(false, std::memory_order_relaxed);
    if (rc == MGPU_OK) rc = wrc;
    if (rc == MGPU_OK) rc = feed_end(c);
    if (rc == MGPU_OK) {
        c->stream_pos += length;
        c->acc.total_ms = (float) (wall_ms() - t_start);
        c->timing = c->acc;
    }
    return rc;
}

// ---- one capture sharded by buffer ranges over several contexts / GPUs (BASELINE config 5) ------------------
// Buffers are independent except for the ICAO filter, and the pre-screen needs the adder addresses of the WHOLE
// capture (a frame is only "conditional" with respect to adds that may lie in an earlier shard).  So a shard runs
// twice: pass 1 (mode 1) sweeps it for its adder bitmap; the bitmaps are OR-ed across shards (the exchange step:
// 2 MiB per rank); pass 2 (mode 2) runs convert, sweep and pre-screen against the global bitmap and keeps every
// chunk's live records as a packet.  The packets of all shards, in stream order, go through mgpu_walk_packets on
// one context: the ordered walk and the message build, exactly as for an unsharded stream.

// A context that starts (or continues) in the middle of a capture: its sample clock, and the 326 magnitudes that precede the first
// sample (sdr_ifile.c:209-213) from the 326 IQ samples before it.
static int start_mid_stream(mgpu_ctx *c, uint64_t first_sample, const void *history_iq) {
    c->stream_pos = first_sample;
    c->eof = false;
    c->tail_src = nullptr;
    if (first_sample) {
        const size_t bps = c->cfg.format == MGPU_FMT_UC8 ? 2 : 4;
        if (!c->d_hist) {
            HIPCHK(c, hipMalloc(&c->d_hist, (2 * kTrailing + 64) * sizeof(uint16_t)));
            HIPCHK(c, hipMalloc(&c->d_hist_iq, kTrailing * 4 + 64));
            HIPCHK(c, hipMalloc(&c->d_hist_sums, 8 * sizeof(unsigned long long)));
        }
        hipStream_t s = c->stream;
        HIPCHK(c, hipMemcpyAsync(c->d_hist_iq, history_iq, kTrailing * bps, hipMemcpyHostToDevice, s));
        HIPCHK(c, hipMemsetAsync(c->d_hist_sums, 0, 8 * sizeof(unsigned long long), s));
        ConvertParams cp{};
        cp.iq = c->d_hist_iq; cp.mag = c->d_hist; cp.n = kTrailing; cp.buf_samples = 0x80000000u;
        cp.uc8_folded = c->d_uc8_folded;
        cp.sum_level = c->d_hist_sums; cp.sum_power = c->d_hist_sums + 1;
        cp.fsum_level = (double *) (c->d_hist_sums + 2); cp.fsum_power = (double *) (c->d_hist_sums + 3);
        launch_convert(c->cfg.format, cp, s);
        HIPCHK(c, hipStreamSynchronize(s));
        c->tail_src = c->d_hist + kTrailing;   // d_hist[326 + i] = magnitude of history sample i
    }
    return MGPU_OK;
}

int mgpu_shard_begin(mgpu_ctx *c, uint64_t first_sample, const void *history_iq, int mode) {
    if (!c || mode < 1 || mode > 3 || first_sample % c->cfg.buf_samples || c->deferred) return MGPU_E_INVAL;
    if (first_sample && !history_iq) return MGPU_E_INVAL;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    c->shard_mode = mode;
    if (mode == 2)                           // the packets carry every live record's would-be skip-window counts
        for (auto &sl : c->slot)
            if (!sl.d_live_win) {
                HIPCHK(c, hipMalloc(&sl.d_live_win, c->cap_pool * sizeof(unsigned long long)));
                HIPCHK(c, hipHostMalloc(&sl.h_live_win, c->cap_pool * sizeof(unsigned long long)));
            }
    c->shard_packets.clear();
    c->shard_est.clear(); c->shard_est_pos.clear(); c->shard_est_off.clear();
    { const int rc = start_mid_stream(c, first_sample, history_iq); if (rc != MGPU_OK) return rc; }
    return MGPU_OK;
}

int mgpu_adder_bitmap_get(mgpu_ctx *c, uint32_t *words) {
    if (!c || !words) return MGPU_E_INVAL;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    HIPCHK(c, hipMemcpy(words, c->d_adder_bitmap, (1u << 24) / 8, hipMemcpyDeviceToHost));
    return MGPU_OK;
}

int mgpu_adder_bitmap_set(mgpu_ctx *c, const uint32_t *words) {
    if (!c || !words) return MGPU_E_INVAL;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    HIPCHK(c, hipMemcpy(c->d_adder_bitmap, words, (1u << 24) / 8, hipMemcpyHostToDevice));
    return MGPU_OK;
}

int mgpu_shard_packets(mgpu_ctx *c, const void **packets, uint64_t *bytes) {
    if (!c || !packets || !bytes) return MGPU_E_INVAL;
    *packets = c->shard_packets.data();
    *bytes = c->shard_packets.size();
    return MGPU_OK;
}

// One packet = one chunk of some rank's range, as fetcher_main lays it out.
struct PacketView {
    uint64_t pos = 0, n = 0, nrecs = 0, nbuf = 0;
    uint64_t hdr[kPacketWords] = {};
    const PhaseRec *recs = nullptr;                            // nrecs records + the walk's sentinel
    const unsigned long long *sig = nullptr, *win = nullptr;    // per record: its would-be signal power, the counts of its would-be skip window
    const unsigned long long *sums = nullptr;                   // level[nbuf], power[nbuf]: integers (UC8) or doubles, eight bytes each
};

// The packets may come from other ranks over a gather: nothing in a header is trusted before it is checked against the bytes
// that are really there (record count without a 64-bit overflow) and the context's capacity; `check_records`: the order the walk
// relies on, record by record (a rank's own packets, made by this library in this process, are taken as they are).
static int parse_packet(mgpu_ctx *c, const uint8_t *&p, const uint8_t *end, PacketView &v, bool check_records) {
    constexpr uint64_t kRecBytes = sizeof(PhaseRec) + 16;       // record + its signal power + its window counts
    if ((size_t) (end - p) < sizeof(v.hdr)) { c->err = "shard packets: truncated packet header"; return MGPU_E_INVAL; }
    std::memcpy(v.hdr, p, sizeof(v.hdr));
    p += sizeof(v.hdr);
    v.pos = v.hdr[0]; v.n = v.hdr[1]; v.nrecs = v.hdr[2]; v.nbuf = v.hdr[10];
    if (v.hdr[3] != kPacketMagic || v.n == 0 || v.n > c->cap_samples || v.n > 0xFFFFFFF0ull ||
        v.nbuf != (v.n + c->cfg.buf_samples - 1) / c->cfg.buf_samples || (uint64_t) (end - p) < sizeof(PhaseRec) ||
        v.nrecs > ((uint64_t) (end - p) - sizeof(PhaseRec)) / kRecBytes ||
        (uint64_t) (end - p) - sizeof(PhaseRec) - v.nrecs * kRecBytes < v.nbuf * 16) {
        c->err = "shard packets: a packet must lie within max_samples, with all its records present";
        return MGPU_E_INVAL;
    }
    // the records are walked where they lie (packets are 8-byte aligned and a sentinel record follows the last one)
    v.recs = (const PhaseRec *) p;
    p += (v.nrecs + 1) * sizeof(PhaseRec);
    v.sig = (const unsigned long long *) p;
    p += v.nrecs * 8;
    v.win = (const unsigned long long *) p;
    p += v.nrecs * 8;
    v.sums = (const unsigned long long *) p;
    p += v.nbuf * 16;
    if (v.recs[v.nrecs].pos != 0xFFFFFFFFu) { c->err = "shard packets: malformed record list"; return MGPU_E_INVAL; }
    if (check_records)
        for (uint64_t i = 0; i < v.nrecs; ++i) {                 // sorted by position, inside the packet's samples, a real phase
            const PhaseRec &r = v.recs[i];
            if (r.pos >= v.n || (i && r.pos < v.recs[i - 1].pos) || r.phase < 4 || r.phase > 8) {
                c->err = "shard packets: malformed record list";
                return MGPU_E_INVAL;
            }
        }
    return MGPU_OK;
}

// A packet's ordered walk (the walker's team, as for a chunk of an unsharded stream); build: its messages appended to
// c->pending and every statistic an unsharded run keeps — the sweep-side tallies and the per-buffer sums ride in the packet,
// what the accepted frames' skip windows hide is the sum of the accepted records' window counts; noise_terms (when given): what
// each buffer adds to noise_power_sum, in order (a double sum is order-dependent: the rank that combines ranges re-adds them).
static int walk_one_packet(mgpu_ctx *c, const PacketView &v, bool build, std::vector<double> *noise_terms) {
    HostJob &job = c->job[0];
    mgpu_counters &k = c->counters;
    const uint64_t nrecs = v.nrecs, nbuf = v.nbuf, n = v.n;
    const PhaseRec *recs = v.recs;
    const unsigned long long *sig = v.sig, *win = v.win, *sums = v.sums;
    ifile_grid(c, v.pos, n, job.buffers);
    const uint64_t cap = nrecs + 1;
    job.pos.resize(cap); c->w_limit.resize(cap); c->w_skip.resize(cap);
    job.rc = ResolveCounts();
    const double t0 = wall_ms();
    const int64_t wn = host_walk(c, job, recs, job.buffers, nrecs, cap);
    c->acc.resolve_ms += (float) (wall_ms() - t0);
    if (wn < 0) return MGPU_E_OVERFLOW;
    if (!build) return MGPU_OK;
    const double t1 = wall_ms();
    const size_t first = c->pending.size();
    if (!c->pending.grow_for((size_t) wn)) return c->pending.external ? MGPU_E_OVERFLOW : MGPU_E_NOMEM;
    {
        mgpu_msg *dst = c->pending.data() + first;
        const int parts = wn >= 4096 ? c->build_threads : 1;
        c->build_team.run(parts, [&](int i) {
            const uint64_t lo = (uint64_t) wn * i / parts, hi = (uint64_t) wn * (i + 1) / parts;
            Resolver::build_messages(recs, sig, nullptr, job.buffers, job.acc.data() + lo, hi - lo, dst + lo);
        });
    }
    c->pending.n = first + (size_t) wn;
    // ---- the statistics: feed_end's and build_job's, from what the packet carries ----
    const ResolveCounts &rc = job.rc;
    for (int i = 0; i < 3; ++i) k.demod_accepted[i] += rc.accepted[i];
    for (int i = 0; i < 5; ++i) k.demod_bestPhase[i] += rc.best_phase[i];
    uint64_t hw[5] = {0, 0, 0, 0, 0};                        // what the accepted frames' skip windows hide: candidates, phases 4/5, 6/7, 8, conditional-only
    std::vector<uint64_t> buf_scaled(nbuf, 0);
    for (int64_t i = 0; i < wn; ++i) {
        const Accepted &a = job.acc[(size_t) i];
        const unsigned long long w = win[a.rec];
        hw[0] += w & 0xff; hw[1] += (w >> 8) & 0xff; hw[2] += (w >> 16) & 0xff; hw[3] += (w >> 24) & 0xff; hw[4] += (w >> 32) & 0xff;
        const unsigned long long sumsq = sig[a.rec];
        const unsigned sig_len = (recs[a.rec].msg[0] & 0x80) ? 268u : 134u;    // msglen * 12 / 5, demod_2400.c:439
        const double signal_power = (double) sumsq / 65535.0 / 65535.0, level = signal_power / sig_len;
        k.signal_power_sum += signal_power;
        k.signal_power_count += sig_len;
        if (level > k.peak_signal_power) k.peak_signal_power = level;
        if (level > 0.50119) k.strong_signal_count++;
        if (a.buffer < nbuf) buf_scaled[a.buffer] += sumsq;
    }
    const uint64_t C = v.hdr[4], U = v.hdr[8], R = v.hdr[9], cW = hw[0], uW = hw[4];
    k.demod_preambles += C - cW;
    k.demod_preamblePhase[0] += v.hdr[5] - hw[1];
    k.demod_preamblePhase[1] += v.hdr[5] - hw[1];
    k.demod_preamblePhase[2] += v.hdr[6] - hw[2];
    k.demod_preamblePhase[3] += v.hdr[6] - hw[2];
    k.demod_preamblePhase[4] += v.hdr[7] - hw[3];
    k.demod_rejected_bad += (C - U - R) - (cW - uW - rc.skipped_uncond_groups) + rc.rejected_bad;
    k.demod_rejected_unknown_icao += rc.rejected_unknown + (U - rc.visited_cond_groups - uW);
    for (uint64_t b = 0; b < nbuf; ++b) {                    // noise power per buffer (demod_2400.c:474-479)
        const BufferClock &bc = job.buffers[b];
        double mean_power;
        if (c->cfg.format == MGPU_FMT_UC8) mean_power = (double) sums[nbuf + b] / 65535.0 / 65535.0 / bc.length;   // convert.c:105-107
        else { double f; std::memcpy(&f, &sums[nbuf + b], 8); mean_power = (double) ((float) f / (float) bc.length); }
        const double term = mean_power * bc.length - (double) buf_scaled[b] / 65535.0 / 65535.0;
        k.noise_power_sum += term;
        if (noise_terms) noise_terms->push_back(term);
        k.noise_power_count += bc.length;
        k.samples_lost += c->cfg.buf_samples - bc.length;    // readsb.c:886
    }
    k.samples_processed += n;
    k.nbuffers += nbuf;
    k.nflips = c->resolver.nflips();
    c->acc.build_ms += (float) (wall_ms() - t1);
    c->acc.n_messages += (uint64_t) wn;
    return MGPU_OK;
}

// Packets that continue the context's stream: per packet the ordered walk, the messages, the statistics.
static int walk_packets_checked(mgpu_ctx *c, const void *packets, uint64_t bytes) {
    const uint8_t *p = (const uint8_t *) packets, *end = p + bytes;
    if ((uintptr_t) packets & 7) { c->err = "mgpu_walk_packets: the packets must be 8-byte aligned"; return MGPU_E_INVAL; }
    while (p < end) {
        PacketView v;
        int rc = parse_packet(c, p, end, v, true);
        if (rc != MGPU_OK) return rc;
        if (v.pos != c->stream_pos) { c->err = "mgpu_walk_packets: packets must continue the stream in order"; return MGPU_E_INVAL; }
        rc = walk_one_packet(c, v, true, nullptr);
        if (rc != MGPU_OK) return rc;
        c->stream_pos += v.n;
        if (v.n % c->cfg.buf_samples) c->eof = true;
    }
    return MGPU_OK;
}

// The context's own packets (one rank holds the whole capture): reset, then walk them where the shard pass left them.
int mgpu_walk_packets(mgpu_ctx *c, const void *packets, uint64_t bytes) {
    if (!c || (!packets && bytes)) return MGPU_E_INVAL;
    if (c->eof) return MGPU_E_EOF;
    if (c->shard_mode != 0 || c->deferred) { c->err = "mgpu_walk_packets: the context is in the middle of a shard pass (or in deferred mode)"; return MGPU_E_INVAL; }
    { std::lock_guard<std::mutex> lk(c->mu); c->hot.store(true, std::memory_order_relaxed); }
    c->cv.notify_all();                      // the walker's team polls instead of sleeping while the packets are walked
    const int rc = guarded(c, [&] { return walk_packets_checked(c, packets, bytes); });
    c->hot.store(false, std::memory_order_relaxed);
    return rc;
}

// ---- config 5 with the ordered walk itself sharded: every rank walks its OWN range (include/modes_gpu.h) ----------------------
//
// What ties the ranges of one capture together is the ICAO filter: its two generations, `occupied`, the table size, and the clock
// of its 60 s expiry — which is data-dependent at millisecond granularity (the expiry after a buffer is tested against the
// timestamp of the buffer's last scored candidate, and the next is due 60 s after THAT: demod_2400.c:412-414, readsb.c:1227-1231),
// so a rank cannot know the schedule from the buffer grid.  The protocol (readsb_amd/shard.py):
//   1. every rank puts warm-up (two filter generations before its range) + range through the GPU pipeline and keeps the packets;
//   2. every rank ESTIMATES its buffers' end clocks from the records alone (mgpu_shard_clock_estimate); all-gather; the schedule
//      is the chain over all end clocks (mgpu_flip_schedule);
//   3. every rank walks warm-up + range with that schedule IMPOSED, from an empty filter at the warm-up's first sample (rank 0:
//      from the reference's initial state), and reports its true end clocks, the state it had at its range's first sample, the
//      state it ended with;
//   4. all-gather; done iff the chain over the true end clocks reproduces the schedule and every rank's state at its first
//      sample equals the state the rank before it ended with.  Otherwise: the new schedule, and a rank whose seam failed starts
//      its range from the imported state of its predecessor instead of its own warm-up; again from 3.
// At the fixed point every rank's walk IS the serial walk's (induction over buffers: rank 0 starts from the true state; the true
// rule expires the filter after buffer b iff the chain says so, because the chain runs the same rule on the same end clocks).
// Every range's messages are built by its own rank; integer counters add up; the two order-dependent double sums are re-added in
// stream order by whoever combines the ranges (mgpu_seqsum*, from the per-buffer terms / the messages themselves).

uint64_t mgpu_flip_schedule(const int64_t *end_clock, uint64_t nbuf, int64_t startup_ms, int filter_clock, uint64_t *flip_after, uint64_t cap) {
    std::vector<uint64_t> f;
    flip_schedule(end_clock, nbuf, startup_ms, filter_clock, f);
    for (size_t i = 0; i < f.size() && i < cap; ++i) flip_after[i] = f[i];
    return f.size();
}

uint64_t mgpu_expiry_windows(uint64_t nbuf_total, uint32_t buf_samples, int64_t startup_ms, int filter_clock, uint8_t *mask) {
    if (!mask && nbuf_total) return 0;
    return expiry_windows(nbuf_total, buf_samples ? buf_samples : 131072u, startup_ms, filter_clock, mask);
}

// What every rank concludes from a round's all-gather — the same on every rank, so no further exchange is needed.
int mgpu_shard_round(const int64_t *sched, uint64_t nsched, uint32_t world, const int64_t *const *clocks, const uint64_t *nclocks,
                     const void *const *state_first, const uint64_t *state_first_bytes, const void *const *state_end, const uint64_t *state_end_bytes,
                     uint64_t nsamples, uint32_t buf_samples, int64_t startup_ms, int filter_clock,
                     int64_t *next_sched, uint64_t cap, uint64_t *n_next, int32_t *import_from, int32_t *done) {
    if (!world || !clocks || !nclocks || !n_next || !import_from || !done || (nsched && !sched)) return MGPU_E_INVAL;
    if (!buf_samples) buf_samples = 131072;
    std::vector<int64_t> all;
    for (uint32_t r = 0; r < world; ++r) all.insert(all.end(), clocks[r], clocks[r] + nclocks[r]);
    if (nsamples % buf_samples == 0) all.push_back((int64_t) ((nsamples * 5) / 12000) + startup_ms);   // the EOF buffer (sdr_ifile.c:223-237): mgpu_finish's clock
    std::vector<uint64_t> f;
    flip_schedule(all.data(), all.size(), startup_ms, filter_clock, f);
    *n_next = f.size();
    bool same = f.size() == nsched;
    for (size_t i = 0; i < f.size(); ++i) {
        const int64_t ts = (int64_t) (f[i] * buf_samples) * 5;
        if (i < cap && next_sched) next_sched[i] = ts;
        if (same && sched[i] != ts) same = false;
    }
    if (f.size() > cap) return MGPU_E_CAPACITY;
    bool seams = true;
    int32_t prev = -1;                                          // the last rank with a range of its own (an empty range passes its neighbour's state through)
    for (uint32_t r = 0; r < world; ++r) {
        import_from[r] = -1;
        if (nclocks[r] == 0) continue;
        if (prev >= 0 && (state_first_bytes[r] != state_end_bytes[prev] || std::memcmp(state_first[r], state_end[prev], (size_t) state_end_bytes[prev]) != 0)) {
            import_from[r] = prev;
            seams = false;
        }
        prev = (int32_t) r;
    }
    *done = same && seams;
    return MGPU_OK;
}

static int shard_packets_span(mgpu_ctx *c, const void *&packets, uint64_t &bytes) {
    if (!packets) { packets = c->shard_packets.data(); bytes = c->shard_packets.size(); }
    if ((uintptr_t) packets & 7) { c->err = "shard packets must be 8-byte aligned"; return MGPU_E_INVAL; }
    return MGPU_OK;
}

int mgpu_shard_clock_estimate(mgpu_ctx *c, const void *packets, uint64_t bytes, uint64_t own_first, int64_t *end_clocks, uint64_t cap, uint64_t *n_out) {
    if (!c || !end_clocks || !n_out) return MGPU_E_INVAL;
    *n_out = 0;
    if (!packets && !c->shard_est_pos.empty()) {              // the context's own packets: the fetcher has estimated them as they came
        { const int rc = wait_all(c); if (rc != MGPU_OK) return rc; }
        size_t k = 0;
        while (k < c->shard_est_pos.size() && c->shard_est_pos[k] < own_first) ++k;
        const size_t off = k < c->shard_est_off.size() ? (size_t) c->shard_est_off[k] : c->shard_est.size();
        const size_t cnt = c->shard_est.size() - off;
        if (cnt > cap) { c->err = "mgpu_shard_clock_estimate: more buffers than the caller's array holds"; return MGPU_E_CAPACITY; }
        std::memcpy(end_clocks, c->shard_est.data() + off, cnt * sizeof(int64_t));
        *n_out = cnt;
        return MGPU_OK;
    }
    { const int rc = shard_packets_span(c, packets, bytes); if (rc != MGPU_OK) return rc; }
    const uint8_t *p = (const uint8_t *) packets, *end = p + bytes;
    std::vector<BufferClock> bufs;
    std::vector<int64_t> clocks;
    while (p < end) {
        PacketView v;
        const int rc = parse_packet(c, p, end, v, false);
        if (rc != MGPU_OK) return rc;
        if (v.pos < own_first) continue;
        ifile_grid(c, v.pos, v.n, bufs);
        estimate_end_clocks(v.recs, v.nrecs, bufs, clocks);
    }
    if (clocks.size() > cap) { c->err = "mgpu_shard_clock_estimate: more buffers than the caller's array holds"; return MGPU_E_CAPACITY; }
    std::memcpy(end_clocks, clocks.data(), clocks.size() * sizeof(int64_t));
    *n_out = clocks.size();
    return MGPU_OK;
}

static int shard_walk_checked(mgpu_ctx *c, const void *packets, uint64_t bytes, const mgpu_shard_walk_args *a, int64_t *end_clocks, uint64_t cap, uint64_t *n_out) {
    const uint8_t *p = (const uint8_t *) packets, *end = p + bytes;
    const double t_all = wall_ms();
    std::vector<PacketView> views;
    while (p < end) {
        PacketView v;
        const int rc = parse_packet(c, p, end, v, a->check_records != 0);
        if (rc != MGPU_OK) return rc;
        views.push_back(v);
    }
    c->pending.clear();
    std::memset(&c->counters, 0, sizeof(c->counters));
    std::memset(&c->acc, 0, sizeof(c->acc));
    c->shard_noise.clear();
    c->shard_sig.clear();
    c->eof = false;
    c->spec_segments = c->spec_batches = 0;
    c->shard_sched.assign(a->flip_after, a->flip_after + a->nflips);
    ShardWalkPlan plan;
    plan.own_first = a->own_first; plan.buf_samples = c->cfg.buf_samples; plan.startup_ms = c->cfg.startup_time_ms; plan.clock_mode = (int) c->cfg.filter_clock;
    plan.sched = c->shard_sched.data(); plan.nsched = c->shard_sched.size();
    plan.start_state = (const uint8_t *) a->start_state; plan.start_state_bytes = a->start_state_bytes;
    ShardWalkOut &out = c->shard_out;
    const char *err = "";
    double t_own = 0;
    const int rc = shard_walk_core(c->resolver, plan, views.size(),
        [&](size_t i, uint64_t &pos, uint64_t &n) { pos = views[i].pos; n = views[i].n; },
        [&](size_t i, bool own) {
            if (own && t_own == 0) { t_own = wall_ms(); c->acc.resolve_ms = 0; }          // (resolve_ms: the own range's walk; d2h_ms below: the warm-up's)
            const int wrc = walk_one_packet(c, views[i], own, own ? &c->shard_noise : nullptr);
            if (wrc == MGPU_OK && own) { c->stream_pos = views[i].pos + views[i].n; if (views[i].n % c->cfg.buf_samples) c->eof = true; }
            return wrc;
        }, out, &err);
    if (rc == -1) { c->err = std::string("mgpu_shard_walk: ") + err; return MGPU_E_INVAL; }
    if (rc != MGPU_OK) return rc;
    c->counters.nflips = c->resolver.nflips();
    if (out.clocks.size() > cap) { c->err = "mgpu_shard_walk: more buffers than the caller's array holds"; return MGPU_E_CAPACITY; }
    std::memcpy(end_clocks, out.clocks.data(), out.clocks.size() * sizeof(int64_t));
    *n_out = out.clocks.size();
    c->acc.d2h_ms = t_own > 0 ? (float) (t_own - t_all) : 0;
    c->acc.total_ms = (float) (wall_ms() - t_all);
    c->timing = c->acc;
    return MGPU_OK;
}

int mgpu_shard_walk(mgpu_ctx *c, const void *packets, uint64_t bytes, const struct mgpu_shard_walk_args *a, int64_t *end_clocks, uint64_t cap, uint64_t *n_out) {
    if (!c || !a || !end_clocks || !n_out || (a->nflips && !a->flip_after) || (a->start_state && !a->start_state_bytes)) return MGPU_E_INVAL;
    *n_out = 0;
    if (c->deferred || c->cfg.mode_ac || c->cfg.filter_clock == MGPU_FILTER_CLOCK_EXTERNAL || a->own_first % c->cfg.buf_samples) {
        c->err = "mgpu_shard_walk: not in deferred mode, not with Mode A/C or an external filter clock; ranges are whole buffers";
        return MGPU_E_INVAL;
    }
    { const int rc = wait_all(c); if (rc != MGPU_OK) return rc; }
    { const int rc = shard_packets_span(c, packets, bytes); if (rc != MGPU_OK) return rc; }
    { std::lock_guard<std::mutex> lk(c->mu); c->hot.store(true, std::memory_order_relaxed); }
    c->cv.notify_all();                      // the walker's team polls instead of sleeping while the packets are walked
    const int rc = guarded(c, [&] { return shard_walk_checked(c, packets, bytes, a, end_clocks, cap, n_out); });
    c->hot.store(false, std::memory_order_relaxed);
    return rc;
}

// ---- the same rank, its pass through the ORDINARY pipeline (walk and build overlapped with the GPU) ----
// mgpu_shard_walk above walks a range's packets after its GPU pass: nothing overlaps, and of a rank's 26 ms for an eighth of the
// one-hour capture 14 were walk and build (profiles/r04_config5_one_hour_emulate8.json).  When the schedule is known BEFORE the pass
// (readsb_amd/shard.py: it is the chain over end clocks of a few buffers around every expiry's possible positions — a pre-pass over
// ~5 % of the capture), warm-up and range go through the pipeline every other stream goes through: begin (cold start or imported
// state, schedule imposed), feed the warm-up, mark, feed the range, end.  Same outputs as mgpu_shard_walk: true end clocks, the
// states at the range's two ends, per-buffer noise terms; messages and counters by mgpu_collect.
int mgpu_shard_stream_begin(mgpu_ctx *c, const struct mgpu_shard_stream_args *a) {
    if (!c || !a || (a->nflips && !a->flip_after) || (a->start_state && !a->start_state_bytes)) return MGPU_E_INVAL;
    if (c->cfg.mode_ac || c->cfg.filter_clock == MGPU_FILTER_CLOCK_EXTERNAL || a->own_first % c->cfg.buf_samples || a->first_sample % c->cfg.buf_samples ||
        a->first_sample > a->own_first || (a->first_sample && !a->history_iq) || (a->start_state && a->first_sample != a->own_first)) {
        c->err = "mgpu_shard_stream_begin: whole-buffer ranges, no Mode A/C, no external filter clock; an imported state starts at the range's first sample";
        return MGPU_E_INVAL;
    }
    { const int rc = mgpu_reset(c); if (rc != MGPU_OK) return rc; }
    HIPCHK(c, hipSetDevice(c->cfg.device));
    { const int rc = start_mid_stream(c, a->first_sample, a->history_iq); if (rc != MGPU_OK) return rc; }
    c->shard_sched.assign(a->flip_after, a->flip_after + a->nflips);
    for (size_t i = 1; i < c->shard_sched.size(); ++i)
        if (c->shard_sched[i] <= c->shard_sched[i - 1]) { c->err = "mgpu_shard_stream_begin: the schedule must be ascending"; return MGPU_E_INVAL; }
    Resolver &res = c->resolver;
    c->shard_stream_cold = false;
    if (a->start_state) {
        if (!res.import_state((const uint8_t *) a->start_state, a->start_state_bytes)) { c->err = "mgpu_shard_stream_begin: not a filter state"; return MGPU_E_INVAL; }
    } else if (a->first_sample == 0) res.reset(c->cfg.startup_time_ms, (int) c->cfg.filter_clock);
    else { res.reset_empty(c->cfg.startup_time_ms); c->shard_stream_cold = true; }
    res.set_schedule(c->shard_sched.data(), c->shard_sched.size());
    c->shard_stream = true;
    c->shard_marked = false;
    c->shard_noise_on = true;                                  // (the builder skips the warm-up's chunks altogether: only the range's buffers log a term)
    c->shard_stream_own_first = a->own_first;
    c->shard_out.clocks.clear(); c->shard_out.state_first.clear(); c->shard_out.state_end.clear();
    c->shard_noise.clear();
    c->shard_sig.clear();
    return MGPU_OK;
}

// What the walker does when the range begins (in stream order: behind the warm-up's last chunk, ahead of the range's first).
static void shard_mark_now(mgpu_ctx *c) {
    Resolver &res = c->resolver;
    if (c->shard_stream_cold) {                                // the expiries before the range, counted from the schedule
        const int64_t ts0 = (int64_t) c->shard_stream_own_first * 5;
        res.set_nflips((uint64_t) (std::lower_bound(c->shard_sched.begin(), c->shard_sched.end(), ts0) - c->shard_sched.begin()) +
                       (c->cfg.filter_clock == MGPU_FILTER_CLOCK_BEFORE_FIRST ? 1u : 0u));
    }
    res.export_state(c->shard_out.state_first);
    res.log_end_clocks(&c->shard_out.clocks);
    c->shard_marked = true;
}

// Between the warm-up's feeds and the range's.  Synchronous feeds: the warm-up has been walked, the range begins here.  Deferred
// feeds (round 5): nothing waits — the warm-up's chunks may still be anywhere in the pipeline, the walker marks the range's begin
// itself when it gets to its first chunk (walk_job), and the range's kernels run while the warm-up is still being walked.  The
// warm-up's statistics are kept out of every accumulator chunk by chunk (fetch_slot, walk_job, build_job), so one accounting
// period covers the whole pass.
int mgpu_shard_stream_mark(mgpu_ctx *c) {
    if (!c || !c->shard_stream) return MGPU_E_INVAL;
    if (c->stream_pos != c->shard_stream_own_first) { c->err = "mgpu_shard_stream_mark: behind the warm-up's feeds, at the range's first sample"; return MGPU_E_INVAL; }
    if (c->deferred) return MGPU_OK;
    { const int rc = drain(c); if (rc != MGPU_OK) return rc; }
    c->pending.clear();                                        // (the warm-up leaves no messages and no statistics; nflips is set at the end)
    std::memset(&c->counters, 0, sizeof(c->counters));
    if (!c->shard_marked) shard_mark_now(c);
    return MGPU_OK;
}

int mgpu_shard_stream_end(mgpu_ctx *c, int64_t *end_clocks, uint64_t cap, uint64_t *n_out) {
    if (!c || !c->shard_stream || !end_clocks || !n_out) return MGPU_E_INVAL;
    *n_out = 0;
    { const int rc = drain(c); if (rc != MGPU_OK) return rc; }
    if (!c->shard_marked) shard_mark_now(c);                   // (an empty range: no chunk of it ever reached the walker)
    c->resolver.log_end_clocks(nullptr);
    c->shard_noise_on = false;
    c->resolver.export_state(c->shard_out.state_end);
    c->counters.nflips = c->resolver.nflips();
    if (c->shard_out.clocks.size() > cap) { c->err = "mgpu_shard_stream_end: more buffers than the caller's array holds"; return MGPU_E_CAPACITY; }
    std::memcpy(end_clocks, c->shard_out.clocks.data(), c->shard_out.clocks.size() * sizeof(int64_t));
    *n_out = c->shard_out.clocks.size();
    return MGPU_OK;
}

int mgpu_shard_state(mgpu_ctx *c, int which, const void **blob, uint64_t *bytes) {
    if (!c || !blob || !bytes || which < 0 || which > 1) return MGPU_E_INVAL;
    const std::vector<uint8_t> &st = which ? c->shard_out.state_end : c->shard_out.state_first;
    *blob = st.data(); *bytes = st.size();
    return MGPU_OK;
}

int mgpu_shard_signal_terms(mgpu_ctx *c, const uint64_t **terms, uint64_t *n) {
    if (!c || !terms || !n) return MGPU_E_INVAL;
    { const int rc = drain(c); if (rc != MGPU_OK) return rc; }
    if (c->cfg.mode_ac) { *terms = nullptr; *n = 0; return MGPU_OK; }      // (Mode A/C replies sit between the messages and carry no power: the message form)
    *terms = c->shard_sig.data(); *n = c->shard_sig.size();
    return MGPU_OK;
}

int mgpu_shard_noise_terms(mgpu_ctx *c, const double **terms, uint64_t *n) {
    if (!c || !terms || !n) return MGPU_E_INVAL;
    *terms = c->shard_noise.data(); *n = c->shard_noise.size();
    return MGPU_OK;
}

// ---- beast wire format (net_io.c:1655-1714) for message records that already are in HBM -----------------------

// What follows the message list — field decode, beast encoder, tracking gate — runs on a stream of its own (stream_aux): these calls
// are synchronous, and on the pipeline's main stream they waited for every chunk a deferred feed had queued there.
static int beast_reserve(mgpu_ctx *c, uint64_t n) {
    if (n > c->beast_cap_msgs) {
        if (c->d_beast_len) (void) hipFree(c->d_beast_len);
        if (c->d_beast_blocks) (void) hipFree(c->d_beast_blocks);
        if (c->d_beast_off) (void) hipFree(c->d_beast_off);
        c->d_beast_len = nullptr; c->d_beast_blocks = nullptr; c->d_beast_off = nullptr; c->beast_cap_msgs = 0;
        const uint64_t want = n + n / 4 + 1024;
        HIPCHK(c, hipMalloc(&c->d_beast_len, want * sizeof(uint16_t)));
        HIPCHK(c, hipMalloc(&c->d_beast_blocks, 2 * (want / kBlock + 2) * sizeof(uint32_t)));              // frame bytes | deferred messages per workgroup
        HIPCHK(c, hipMalloc(&c->d_beast_off, 2 * (want / kBlock + 2) * sizeof(unsigned long long)));
        c->beast_cap_msgs = want;
    }
    if (!c->d_beast_total) HIPCHK(c, hipMalloc(&c->d_beast_total, 2 * sizeof(unsigned long long)));
    return MGPU_OK;
}

// d_verdict == nullptr: every message's frame.  Everything in device memory; *ndeferred (may be null without a verdict)
static int beast_encode_dev(mgpu_ctx *c, const mgpu_msg *d_msgs, const uint8_t *d_verdict, uint64_t n, uint32_t flags, uint8_t *d_out, uint64_t cap,
                            uint64_t *bytes, mgpu_deferred *d_deferred, uint64_t deferred_cap, uint64_t *ndeferred) {
    if (int rc = beast_reserve(c, n)) return rc;
    const size_t nb = (size_t) (c->beast_cap_msgs / kBlock + 2);
    launch_beast_encode(d_msgs, n, c->d_beast_len, c->d_beast_blocks, c->d_beast_off, d_out, cap, c->d_beast_total, c->stream_aux, d_verdict,
                        (flags & MGPU_BEAST_NET_RULE) ? 1 : 0, c->d_beast_blocks + nb, c->d_beast_off + nb, d_deferred, deferred_cap);
    HIPCHK(c, hipGetLastError());
    unsigned long long total[2] = {0, 0};
    HIPCHK(c, hipMemcpyAsync(total, c->d_beast_total, (d_verdict ? 2 : 1) * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream_aux));
    HIPCHK(c, hipStreamSynchronize(c->stream_aux));
    *bytes = total[0];
    if (ndeferred) *ndeferred = total[1];
    if (total[0] > cap) { c->err = "mgpu_beast_encode: output buffer too small"; return MGPU_E_OVERFLOW; }
    if (d_verdict && total[1] > deferred_cap) { c->err = "mgpu_beast_encode_gated: more deferred messages than the list holds"; return MGPU_E_OVERFLOW; }
    return MGPU_OK;
}

int mgpu_beast_encode_device(mgpu_ctx *c, const struct mgpu_msg *d_msgs, uint64_t n, uint8_t *d_out, uint64_t cap, uint64_t *bytes) {
    if (!c || !bytes || (n && (!d_msgs || !d_out))) return MGPU_E_INVAL;
    *bytes = 0;
    if (n == 0) return MGPU_OK;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    return beast_encode_dev(c, d_msgs, nullptr, n, 0, d_out, cap, bytes, nullptr, 0, nullptr);
}

int mgpu_beast_encode_gated_device(mgpu_ctx *c, const struct mgpu_msg *d_msgs, const uint8_t *d_verdict, uint64_t n, uint32_t flags, uint8_t *d_out,
                                   uint64_t cap, uint64_t *bytes, struct mgpu_deferred *d_deferred, uint64_t deferred_cap, uint64_t *ndeferred) {
    if (!c || !bytes || !ndeferred || (n && (!d_msgs || !d_verdict || !d_out)) || (deferred_cap && !d_deferred)) return MGPU_E_INVAL;
    *bytes = 0; *ndeferred = 0;
    if (n == 0) return MGPU_OK;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    return beast_encode_dev(c, d_msgs, d_verdict, n, flags, d_out, cap, bytes, d_deferred, deferred_cap, ndeferred);
}

static int stage_messages(mgpu_ctx *c, const struct mgpu_msg *msgs, uint64_t n) {          // host list -> d_beast_in
    if (n * sizeof(mgpu_msg) > c->beast_cap_in) {
        if (c->d_beast_in) (void) hipFree(c->d_beast_in);
        c->d_beast_in = nullptr; c->beast_cap_in = 0;
        const uint64_t want = (n + n / 4 + 1024) * sizeof(mgpu_msg);
        HIPCHK(c, hipMalloc(&c->d_beast_in, want));
        c->beast_cap_in = want;
    }
    HIPCHK(c, hipMemcpyAsync(c->d_beast_in, msgs, n * sizeof(mgpu_msg), hipMemcpyHostToDevice, c->stream_aux));
    return MGPU_OK;
}

static int reserve_beast_out(mgpu_ctx *c, uint64_t cap) {
    if (cap > c->beast_cap_out) {
        if (c->d_beast_out) (void) hipFree(c->d_beast_out);
        c->d_beast_out = nullptr; c->beast_cap_out = 0;
        HIPCHK(c, hipMalloc(&c->d_beast_out, cap + 64));
        c->beast_cap_out = cap;
    }
    return MGPU_OK;
}

int mgpu_beast_encode(mgpu_ctx *c, const struct mgpu_msg *msgs, uint64_t n, uint8_t *out, uint64_t cap, uint64_t *bytes) {
    if (!c || !bytes || (n && (!msgs || !out))) return MGPU_E_INVAL;
    *bytes = 0;
    if (n == 0) return MGPU_OK;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    if (int rc = stage_messages(c, msgs, n)) return rc;
    if (int rc = reserve_beast_out(c, cap)) return rc;
    const int rc = beast_encode_dev(c, (const mgpu_msg *) c->d_beast_in, nullptr, n, 0, c->d_beast_out, cap, bytes, nullptr, 0, nullptr);
    if (rc != MGPU_OK) return rc;
    HIPCHK(c, hipMemcpy(out, c->d_beast_out, *bytes, hipMemcpyDeviceToHost));
    return MGPU_OK;
}

// ---- per-message field decode (mode_s.c:598-760, 806-1555; mode_ac.c:171-200) --------------------------------------

static int fields_tables(mgpu_ctx *c) {
    if (c->d_roll_tan) return MGPU_OK;
    const std::vector<double> t = build_roll_tangent_table();
    HIPCHK(c, hipMalloc(&c->d_roll_tan, t.size() * sizeof(double)));
    HIPCHK(c, hipMemcpy(c->d_roll_tan, t.data(), t.size() * sizeof(double), hipMemcpyHostToDevice));
    return MGPU_OK;
}

int mgpu_decode_fields_device(mgpu_ctx *c, const struct mgpu_msg *d_msgs, uint64_t n, struct mgpu_fields *d_out) {
    if (!c || (n && (!d_msgs || !d_out))) return MGPU_E_INVAL;
    if (n == 0) return MGPU_OK;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    if (int rc = fields_tables(c)) return rc;
    launch_decode_fields(d_msgs, n, d_out, c->d_roll_tan, c->stream_aux);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream_aux));
    return MGPU_OK;
}

static int fields_reserve(mgpu_ctx *c, uint64_t n) {
    if (n > c->fields_cap) {
        if (c->d_fields) (void) hipFree(c->d_fields);
        c->d_fields = nullptr; c->fields_cap = 0;
        const uint64_t want = n + n / 4 + 1024;
        HIPCHK(c, hipMalloc(&c->d_fields, want * sizeof(mgpu_fields)));
        c->fields_cap = want;
    }
    return fields_tables(c);
}

int mgpu_decode_fields(mgpu_ctx *c, const struct mgpu_msg *msgs, uint64_t n, struct mgpu_fields *out) {
    if (!c || (n && (!msgs || !out))) return MGPU_E_INVAL;
    if (n == 0) return MGPU_OK;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    if (int rc = fields_reserve(c, n)) return rc;
    if (int rc = stage_messages(c, msgs, n)) return rc;
    launch_decode_fields((const mgpu_msg *) c->d_beast_in, n, c->d_fields, c->d_roll_tan, c->stream_aux);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(out, c->d_fields, n * sizeof(mgpu_fields), hipMemcpyDeviceToHost, c->stream_aux));
    HIPCHK(c, hipStreamSynchronize(c->stream_aux));
    return MGPU_OK;
}

// ---- first stage of the tracker + forwarding rule (track.c:1688-1693, 1905-1966; net_io.c:5846-5849, 5924-5940), kernels/gate.inc ----

static int gate_reserve(mgpu_ctx *c, uint64_t n) {
    if (!c->d_gate_table) {
        HIPCHK(c, hipMalloc(&c->d_gate_table, gate_table_bytes()));
        HIPCHK(c, hipMemsetAsync(c->d_gate_table, 0, gate_table_bytes(), c->stream_aux));
    }
    if (n > c->gate_cap) {
        if (c->d_gate_scratch) (void) hipFree(c->d_gate_scratch);
        if (c->d_gate_verdict) (void) hipFree(c->d_gate_verdict);
        c->d_gate_scratch = nullptr; c->d_gate_verdict = nullptr; c->gate_cap = 0;
        const uint64_t want = n + n / 4 + 1024;
        HIPCHK(c, hipMalloc(&c->d_gate_scratch, gate_scratch_bytes(want)));
        HIPCHK(c, hipMalloc(&c->d_gate_verdict, want));
        c->gate_cap = want;
    }
    return MGPU_OK;
}

int mgpu_track_gate_reset(mgpu_ctx *c) {
    if (!c) return MGPU_E_INVAL;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    if (c->d_gate_table) {
        HIPCHK(c, hipMemsetAsync(c->d_gate_table, 0, gate_table_bytes(), c->stream_aux));
        HIPCHK(c, hipStreamSynchronize(c->stream_aux));
    }
    return MGPU_OK;
}

int mgpu_track_gate_device(mgpu_ctx *c, const struct mgpu_msg *d_msgs, const struct mgpu_fields *d_fields, uint64_t n, uint8_t *d_verdict) {
    if (!c || (n && (!d_msgs || !d_fields || !d_verdict)) || n > 0xffffffffull) return MGPU_E_INVAL;
    if (n == 0) return MGPU_OK;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    if (int rc = gate_reserve(c, n)) return rc;
    launch_track_gate(d_msgs, d_fields, n, c->cfg.buf_samples, c->d_gate_table, c->d_gate_scratch, d_verdict, c->stream_aux);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream_aux));
    return MGPU_OK;
}

// host list -> d_beast_in, its field records -> d_fields, its verdicts (continuing the context's aircraft table) -> d_gate_verdict
static int gate_staged(mgpu_ctx *c, const struct mgpu_msg *msgs, uint64_t n) {
    if (int rc = fields_reserve(c, n)) return rc;
    if (int rc = gate_reserve(c, n)) return rc;
    if (int rc = stage_messages(c, msgs, n)) return rc;
    launch_decode_fields((const mgpu_msg *) c->d_beast_in, n, c->d_fields, c->d_roll_tan, c->stream_aux);
    launch_track_gate((const mgpu_msg *) c->d_beast_in, c->d_fields, n, c->cfg.buf_samples, c->d_gate_table, c->d_gate_scratch, c->d_gate_verdict, c->stream_aux);
    HIPCHK(c, hipGetLastError());
    return MGPU_OK;
}

int mgpu_track_gate(mgpu_ctx *c, const struct mgpu_msg *msgs, uint64_t n, uint8_t *verdict) {
    if (!c || (n && (!msgs || !verdict)) || n > 0xffffffffull) return MGPU_E_INVAL;
    if (n == 0) return MGPU_OK;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    if (int rc = gate_staged(c, msgs, n)) return rc;
    HIPCHK(c, hipMemcpyAsync(verdict, c->d_gate_verdict, n, hipMemcpyDeviceToHost, c->stream_aux));
    HIPCHK(c, hipStreamSynchronize(c->stream_aux));
    return MGPU_OK;
}

// The gate's verdict applied to the encoder: the beast stream of what the reference forwards for certain + the list of the
// messages its position tracker has to settle (include/modes_gpu.h).  Host arrays; the aircraft table goes on from call to call.
int mgpu_beast_encode_gated(mgpu_ctx *c, const struct mgpu_msg *msgs, uint64_t n, uint32_t flags, uint8_t *out, uint64_t cap, uint64_t *bytes,
                            struct mgpu_deferred *deferred, uint64_t deferred_cap, uint64_t *ndeferred) {
    if (!c || !bytes || !ndeferred || (n && (!msgs || !out)) || (deferred_cap && !deferred) || n > 0xffffffffull) return MGPU_E_INVAL;
    *bytes = 0; *ndeferred = 0;
    if (n == 0) return MGPU_OK;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    if (int rc = gate_staged(c, msgs, n)) return rc;
    if (int rc = reserve_beast_out(c, cap)) return rc;
    if (deferred_cap > c->deferred_cap) {
        if (c->d_deferred) (void) hipFree(c->d_deferred);
        c->d_deferred = nullptr; c->deferred_cap = 0;
        HIPCHK(c, hipMalloc(&c->d_deferred, (deferred_cap + 64) * sizeof(mgpu_deferred)));
        c->deferred_cap = deferred_cap + 64;
    }
    const int rc = beast_encode_dev(c, (const mgpu_msg *) c->d_beast_in, c->d_gate_verdict, n, flags, c->d_beast_out, cap, bytes, c->d_deferred, deferred_cap, ndeferred);
    if (rc != MGPU_OK) return rc;
    HIPCHK(c, hipMemcpy(out, c->d_beast_out, *bytes, hipMemcpyDeviceToHost));
    if (*ndeferred) HIPCHK(c, hipMemcpy(deferred, c->d_deferred, *ndeferred * sizeof(mgpu_deferred), hipMemcpyDeviceToHost));
    return MGPU_OK;
}

uint32_t mgpu_crc_checksum(const uint8_t *msg, int bits) { return crc_tables().checksum(msg, bits); }

static const std::vector<SyndromeEntry> &host_table(int nfix, int bits) {
    static std::vector<SyndromeEntry> cache[3][2];
    static bool built[3][2];
    const int n = nfix < 0 ? 0 : nfix > 2 ? 2 : nfix, b = bits == 56 ? 0 : 1;
    if (!built[n][b]) { cache[n][b] = build_syndrome_table(bits == 56 ? 56 : 112, n); built[n][b] = true; }
    return cache[n][b];
}

int mgpu_crc_diagnose(int nfix_crc, uint32_t syndrome, int bits, int *bit0, int *bit1) {
    if (bit0) *bit0 = -1;
    if (bit1) *bit1 = -1;
    if (syndrome == 0) return 0;
    const std::vector<SyndromeEntry> &t = host_table(nfix_crc, bits);
    size_t lo = 0, hi = t.size();
    while (lo < hi) {
        size_t mid = (lo + hi) / 2;
        if (t[mid].syndrome < syndrome) lo = mid + 1; else hi = mid;
    }
    if (lo == t.size() || t[lo].syndrome != syndrome) return -1;
    if (bit0) *bit0 = t[lo].bit0;
    if (bit1 && t[lo].nerr > 1) *bit1 = t[lo].bit1;
    return t[lo].nerr;
}

int mgpu_crc_table_size(int nfix_crc, int bits) { return (int) host_table(nfix_crc, bits).size(); }

const uint16_t *mgpu_uc8_table(void) { return uc8_table(); }

}  // extern "C"
