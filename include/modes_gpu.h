/* modes_gpu.h — C ABI of libmodes_gpu.so, the MI355X (gfx950) implementation of readsb's
 * Mode-S demodulator hot path:
 *
 *     IQ bytes -> magnitude (convert.c) -> 2.4 MSps preamble sweep + PPM bit slicer
 *     (demod_2400.c demodulate2400) -> CRC-24 + 1/2-bit repair (crc.c) -> scored, ordered
 *     message records ready for decodeModesMessage()/track.c.
 *
 * Everything here is `extern "C"`, plain pointers and sizes; the caller owns all host
 * memory, the library owns all device memory.  One context = one SDR stream on one GPU;
 * calls into a context must come from one thread at a time (the reference calls demodulate2400
 * from one decode thread only, readsb.c:871); inside, a context runs its own host threads
 * (mgpu_host_cpus).  Each entry point names the reference interface it replaces
 * (file:line under the reference tree).
 *
 * The reference-side binding a maintainer would add is shown in INTEGRATION.md.
 */
#ifndef MODES_GPU_H
#define MODES_GPU_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* input_format_t, convert.h:28-31 (same numeric values) */
#define MGPU_FMT_UC8     0
#define MGPU_FMT_SC16    1
#define MGPU_FMT_SC16Q11 2

/* status codes (0 = ok, negative = error; text via mgpu_strerror) */
#define MGPU_OK              0
#define MGPU_E_INVAL        -1   /* bad argument / bad state */
#define MGPU_E_NODEVICE     -2   /* no usable HIP device: the library never falls back to the CPU */
#define MGPU_E_HIP          -3   /* a HIP runtime call failed (message via mgpu_last_error) */
#define MGPU_E_NOMEM        -4
#define MGPU_E_OVERFLOW     -5   /* a device-side pool was too small; recreate with a larger pool */
#define MGPU_E_CAPACITY     -6   /* more samples than cfg.max_samples in one call */
#define MGPU_E_EOF          -7   /* stream already ended on a short buffer (sdr_ifile.c:223-237) */

/* Bumped whenever a struct of this header grows or an entry point changes meaning.  5 (round 5): mgpu_config.abi_version is checked
 * by mgpu_create, mgpu_abi_version() exported; 4 (round 4): mgpu_config.chunk_buffers; with MGPU_FILTER_CLOCK_EXTERNAL the filter
 * starts EMPTY since 3 (the host mirrors modesInit's icaoFilterAdd(Modes.show_only)). */
#define MGPU_ABI_VERSION 6
/* The version the loaded library was built with: a host compares it with the MGPU_ABI_VERSION of the header it was compiled against. */
uint32_t mgpu_abi_version(void);

typedef struct mgpu_ctx mgpu_ctx;

/* Options of the hot path, named after the struct _Modes fields they mirror. */
struct mgpu_config {
    int32_t  device;              /* HIP device ordinal */
    int32_t  format;              /* MGPU_FMT_*: --iformat (sdr_ifile.c:82-108) */
    int32_t  nfix_crc;            /* Modes.nfix_crc: 0 --no-fix, 1 --fix (default, readsb.c:150), 2 --aggressive */
    int32_t  fixDF;               /* Modes.fixDF (readsb.c:194), 0 with --no-fix-df */
    int32_t  preamble_threshold;  /* Modes.preambleThreshold, default 58 (readsb.c:2268); 1 .. 4095 (the reference clamps to 40 .. 400,
                                   * readsb.c:1473; beyond 6000 k_sweep's 32-bit accumulators could overflow): MGPU_E_INVAL otherwise */
    uint32_t buf_samples;         /* Modes.sdr_buf_samples, default 131072 (readsb.c:2212); multiple of 4096 */
    uint32_t trailing_samples;    /* Modes.trailing_samples = 326 (readsb.c:288); must be 326 */
    uint32_t mode_ac;             /* Modes.mode_ac (--modeac): also run demodulate2400AC on every buffer, readsb.c:871-874.  Every
                                   * format: its noise floor comes from the buffer's mean level / power, which for SC16 / SC16Q11 are
                                   * the reference's sequential float sums (convert.c:225-249), reproduced bit for bit. */
    uint64_t max_samples;         /* largest number of new samples one mgpu_feed_* call may carry */
    int64_t  startup_time_ms;     /* Modes.startup_time: wall clock (ms) the 12 MHz sample clock is anchored to */
    uint64_t record_pool_records; /* device pool for per-phase candidate records; 0 = max_samples/16 + 65536 */
    uint64_t max_messages;        /* cap on accepted messages kept per feed; 0 = max_samples/64 + 65536 */
    uint32_t filter_clock;        /* MGPU_FILTER_CLOCK_*: who runs icaoFilterExpire (readsb.c:1227-1231), see below */
    uint32_t streams_on_device;   /* how many contexts this process runs on the device (0 / 1: this one alone).  With several, every one of them
                                   * keeps its host stages small (4 walkers, 3 builders): the stage threads poll while a feed runs, and eight
                                   * contexts with a lone stream's teams (8 + 6) oversubscribe the device's two L3 groups — measured, 8 streams:
                                   * 9.4 Gsamples/s, with small teams 18.7 (profiles/r03_fanin.txt) */
    uint32_t chunk_buffers;       /* buffers per pipeline chunk (one launch of every kernel); 0 = 1024 (half as many kernel boundaries and tails
                                   * as 512: 325 against 300 Gsamples/s, profiles/r04_chunk_buffers.txt); a host that wants its messages sooner
                                   * takes fewer (latency = three chunks), one that wants throughput takes 2048 and keeps two feeds of 4096
                                   * buffers enqueued ahead of the one it collects (1.12 against 1.21 ms per 537 M samples: the kernels'
                                   * ramps and tails again; the host's ordered walk takes a chunk in rounds of 1024 buffers whatever its
                                   * length, profiles/r06_chunk_2048.txt; memory per context doubles with it: ten slots of ~3 GB) */
    uint32_t abi_version;         /* MGPU_ABI_VERSION of the header the HOST was compiled against (mgpu_config_defaults below passes it);
                                   * mgpu_create() refuses another (MGPU_E_INVAL) — a host built against an older, shorter struct would
                                   * otherwise have the bytes behind its struct read as chunk_buffers.  (The last field: it sits where such a
                                   * host's struct has ended.) */
};

/* The ICAO filter's 60 s expiry (backgroundTasks, readsb.c:1227-1231: `static next_flip = 0`, so the first
 * icaoFilterExpire() runs at the first backgroundTasks call).  The reference program itself has two start-up orders:
 * its decode thread either finds the first buffer waiting (flip AFTER buffer 0, next one 60 s after the clock at the end
 * of buffer 0) or does not (flip BEFORE buffer 0, on the still empty filter, next one 60 s after start-up) — thread
 * scheduling decides (readsb.c:857-902).  The orders differ in which filter generation buffer 0's addresses land in, and
 * therefore in what the next table resize drops (icao_filter.c:65-93).
 *   AFTER_FIRST   the library runs the clock on the buffers' sysTimestamp, first expiry after buffer 0 (default; what the
 *                 oracle, the goldens and a reference run with a slow decode-thread start do)
 *   BEFORE_FIRST  the same clock, first expiry before buffer 0 (anchored to startup_time_ms)
 *   EXTERNAL      the library never expires on its own: the host forwards every icaoFilterExpire() it performs with
 *                 mgpu_filter_expire() (and foreign icaoFilterAdd()s, e.g. from network input, with mgpu_filter_add())
 *                 between two feed calls.  This is what a readsb process linked against the library uses
 *                 (readsb_amd/host/readsb_tree/demod_gpu_wrap.c wraps icaoFilterExpire): its own filter and the
 *                 library's then flip at the same buffer boundaries whatever clock the host runs (synthetic or wall).
 *                 In this mode the filter starts EMPTY: the host's own first add — modesInit's icaoFilterAdd(Modes.show_only),
 *                 readsb.c:310, whatever --show-only is — must be mirrored with mgpu_filter_add() like every other one. */
#define MGPU_FILTER_CLOCK_AFTER_FIRST  0
#define MGPU_FILTER_CLOCK_BEFORE_FIRST 1
#define MGPU_FILTER_CLOCK_EXTERNAL     2

/* Fills cfg with the reference defaults (configSetDefaults, readsb.c:150-228). */
/* Defaults of every field.  C / C++ hosts get the macro: it hands the library the size and the version of the struct THEY were compiled
 * with, the library writes no byte beyond that size and records the version for mgpu_create's check.  The plain entry point writes
 * sizeof(the LIBRARY's struct) bytes and the library's own version: safe only for a caller whose struct is this header revision's —
 * a binding that looks symbols up at run time should carry its own version constant and call mgpu_config_defaults_abi with it, as
 * readsb_amd/binding.py does (ABI_VERSION; it also refuses a library whose mgpu_abi_version() differs). */
void mgpu_config_defaults_abi(struct mgpu_config *cfg, uint32_t struct_bytes, uint32_t abi_version);
void mgpu_config_defaults(struct mgpu_config *cfg);
#ifndef MGPU_NO_DEFAULTS_MACRO
#define mgpu_config_defaults(cfg) mgpu_config_defaults_abi((cfg), (uint32_t) sizeof(struct mgpu_config), MGPU_ABI_VERSION)
#endif

/* One accepted message, in stream order: everything demodulate2400 hands to
 * decodeModesMessage()/netUseMessage() (demod_2400.c:401-471).  64 bytes. */
struct mgpu_msg {
    int64_t  timestamp;       /* mm->timestamp, 12 MHz ticks, end of bit 56 (demod_2400.c:406) */
    int64_t  sysTimestamp;    /* mm->sysTimestamp, ms (demod_2400.c:409) */
    uint64_t sig_sumsq;       /* sum of mag^2 over the signal window (demod_2400.c:442-445) */
    uint16_t sig_len;         /* samples in that window: msglen*12/5 = 268 or 134 */
    int16_t  score;           /* mm->score (scoreModesMessage, mode_s.c:309) */
    uint8_t  phase;           /* bestphase 4..8 */
    uint8_t  correctedbits;   /* mm->correctedbits after decodeModesMessage */
    uint8_t  msgtype;         /* DF after DF repair (mm->msgtype) */
    uint8_t  msgbits;         /* 56 or 112 after DF repair (mm->msgbits) */
    uint32_t addr;            /* address the CRC stage settled on (AA, or the AP syndrome) */
    uint8_t  msg[14];         /* corrected frame = mm->msg after decodeModesMessage */
    uint8_t  raw[14];         /* frame as sliced = what demodulate2400 copies into mm->msg (:420) */
};
/* signalLevel exactly as demod_2400.c:447-448 */
static inline double mgpu_msg_signal_level(const struct mgpu_msg *m) {
    return (double) m->sig_sumsq / 65535.0 / 65535.0 / (double) m->sig_len;
}

/* struct stats demod counters the reference updates inside demodulate2400
 * (stats.h:62-82; demod_2400.c:216,387-456,474-479), accumulated since create/reset. */
struct mgpu_counters {
    uint64_t demod_preambles;
    uint64_t demod_rejected_bad;
    uint64_t demod_rejected_unknown_icao;
    uint64_t demod_accepted[3];
    uint64_t demod_preamblePhase[5];
    uint64_t demod_bestPhase[5];
    uint64_t strong_signal_count;
    uint64_t signal_power_count;
    uint64_t noise_power_count;
    uint64_t samples_processed;
    uint64_t samples_lost;
    uint64_t nbuffers;
    uint64_t nflips;             /* icaoFilterExpire() calls made by the filter clock */
    double   signal_power_sum;
    double   noise_power_sum;
    double   peak_signal_power;
    uint64_t demod_modeac;       /* Mode A/C replies accepted (stats.h:72), only with cfg.mode_ac */
};

/* Where the last feed's time went (ms).  The four kernel figures are pairs of HIP events on the context's main stream around ONE
 * kernel / one group of kernels, on the first chunk since the figures were last reset and every fifteenth after it (seventh until round 6; n_timed_chunks: a timing event costs ~5 us of idle stream; neighbouring figures share an event); a pair adds
 * a constant ~4 us to what it brackets (mgpu_event_bracket_us measures it).  The host figures are wall-clock sums of stage threads
 * that run beside each other and beside the GPU: they overlap, they do not add up to total_ms. */
struct mgpu_timing {
    float h2d_ms;        /* host time spent issuing the IQ block's host->device copies (0 for resident input) */
    float convert_ms;    /* k_convert_* (UC8 / SC16 / SC16Q11 -> magnitudes, per-buffer sums) */
    float sweep_ms;      /* k_sweep alone: the preamble sweep, the kernel the HBM roofline applies to */
    float prescreen_ms;  /* the post-sweep passes: k_count (+ class bitmap) + k_prescreen_write + k_publish */
    float resolve_ms;    /* walker team: the ordered accept / skip-ahead / ICAO filter walk (host wall time) */
    float sigpower_ms;   /* walker thread: launching what follows the walk on the second stream (k_stage_in, k_window_stats incl. the signal powers, k_build_messages) */
    float d2h_ms;        /* fetcher thread: waiting for the chunk, the live records HBM -> page-locked memory -> the walk's memory (host wall time) */
    float total_ms;      /* wall time of the whole call (deferred feeds: since the accounting was opened) */
    uint64_t n_candidates;   /* positions that passed a preamble threshold */
    uint64_t n_records;      /* per-phase records the slicer emitted */
    uint64_t n_live_records; /* records that survived the pre-screen (reach the ordered walk) */
    uint64_t n_messages;     /* accepted messages */
    uint64_t n_chunks;       /* pipeline chunks = launches of each kernel */
    float slice_ms;          /* k_slice: bit slicer + CRC-24 + the filter-independent half of the scoring */
    float build_ms;          /* builder team: struct modesMessage fields + signal / noise statistics — the stage's own work (host wall time;
                              * until ABI 6 this figure also held the wait below) */
    uint64_t n_timed_chunks; /* chunks that carried the stage timing events: convert_ms, sweep_ms, slice_ms and prescreen_ms are sums
                              * over THESE (every 15th chunk; experiments build: MGPU_TIMING_EVERY) */
    float build_wait_ms;     /* builder thread: waiting for the chunk's signal powers (second stream) / the SC16 formats' float sums: idle, not work (ABI 6) */
    float sweep_fused_chunks; /* of the timed chunks, those whose sweep_ms is k_sweep_uc8's — converter and sweep in one kernel (UC8 without Mode A/C):
                               * convert_ms holds nothing for them, the kernel reads the IQ samples and writes the magnitudes (4 B per sample) */
};

/* ---- lifecycle -------------------------------------------------------------------- */

/* Replaces modesInit()'s hot-path part: modesChecksumInit(nfix_crc), icaoFilterInit(),
 * icaoFilterAdd(show_only), init_converter() (readsb.c:306-310, sdr_ifile.c:156).
 * The filter starts holding show_only's default (0xff123456, readsb.h:296) — EXCEPT with cfg.filter_clock ==
 * MGPU_FILTER_CLOCK_EXTERNAL, where it starts EMPTY and the host mirrors its own icaoFilterAdd(Modes.show_only) with mgpu_filter_add
 * like every other add: with a user --show-only a default entry added here as well would leave `occupied` one ahead of the host's
 * and the table resizes (icao_filter.c:65-93) on different adds (readsb_tree/demod_gpu_wrap.c replays the host's early adds). */
int  mgpu_create(const struct mgpu_config *cfg, mgpu_ctx **out);
void mgpu_destroy(mgpu_ctx *ctx);
/* Back to the state right after mgpu_create (sample clock 0, filter = {show_only's default}; empty with the EXTERNAL clock). */
int  mgpu_reset(mgpu_ctx *ctx);
const char *mgpu_strerror(int code);
const char *mgpu_last_error(mgpu_ctx *ctx);
/* 1 when a gfx950-class HIP device is visible, 0 otherwise (never throws). */
int  mgpu_device_count(void);

/* ---- whole hot path: what sdr_ifile.c's reader + the decode thread do per block ------ */

/* ifileRun's read+convert (sdr_ifile.c:194-259) followed by demodulate2400() per
 * 131072-sample buffer (readsb.c:871) and the per-buffer filter clock (readsb.c:1227-1231),
 * for `nsamples` new IQ samples continuing the stream.  nsamples need not be a multiple of
 * buf_samples; a short last buffer ends the stream exactly like a short read() does.
 * Synchronous (unless mgpu_set_deferred): on return the accepted messages are available to mgpu_collect(). */
int mgpu_feed_iq(mgpu_ctx *ctx, const void *iq_host, uint64_t nsamples);

/* Deferred feeds — a continuous stream handed over block after block without the pipeline running empty in between.
 * With on != 0, mgpu_feed_iq / mgpu_feed_iq_device return as soon as the block's chunks are enqueued (they still block while
 * all three pipeline slots are busy), the next block may follow at once, and mgpu_collect() waits for the OLDEST uncollected
 * feed only and returns exactly that feed's messages — so the caller's loop is  feed(k+1); collect(k).  At most 4 feeds may be
 * uncollected.  mgpu_set_message_buffer() then names the array of the NEXT feed (alternate two arrays).  The demod counters
 * settle when nothing is in flight: passing `counters` to mgpu_collect, and mgpu_finish / mgpu_last_timing / mgpu_reset /
 * mgpu_filter_* / mgpu_set_deferred, wait for everything enqueued so far (they "drain"); mgpu_last_timing then covers everything
 * since the previous drain.  The result — messages, their order, every counter — is identical to feeding the same blocks
 * synchronously.  The struct mag_buf entries and the shard calls are refused (MGPU_E_INVAL) while deferred mode is on.
 * Lifetime of the caller's block: a deferred mgpu_feed_iq returns when the last byte of `iq_host` has been copied to the device
 * (the kernels are still to run), so the block may be reused at once, exactly as after a synchronous feed; a device block
 * (mgpu_feed_iq_device) is read by kernels that are still to run and must stay untouched until that feed has been collected.
 * While feeds are in flight the CALLING thread's waits (for a free slot in a feed call, for the oldest feed in mgpu_collect) poll —
 * pauses and sched_yield, as the stage threads' do — instead of sleeping on a condition variable: a wake-up that comes a
 * millisecond late on a loaded host is as long as the work the GPU still has queued. */
int mgpu_set_deferred(mgpu_ctx *ctx, int on);

/* Device-resident messages (deferred mode only, no Mode A/C): the accepted frames' records are built on the GPU
 * (k_build_messages: demod_2400.c:399-445 + the CRC stage's repair, mode_s.c:443-606, lane = message) from the walk's accept
 * list and the live records that already are in HBM, into a list per feed; the host builds nothing and no message crosses PCIe
 * unless asked for.  mgpu_collect_device() waits for the oldest uncollected feed like mgpu_collect() and returns the device
 * pointer of its `*n` records, in stream order — valid until three more feeds have been started — ready for
 * mgpu_decode_fields_device / mgpu_beast_encode_device or an aggregator's RCCL gather (readsb_amd/gather.py: submit_device).
 * mgpu_collect() in this mode copies the whole feed to the host (MGPU_E_OVERFLOW if `cap` is smaller than the feed).
 * The list of a feed holds (chunks per feed) x cfg.max_messages records (default per chunk: samples / 64 + 65536).
 * on == 2 (round 6): built on the GPU the same way, but k_build_messages stores the records straight into the array
 * mgpu_set_message_buffer() named for the feed — which must be page-locked (mgpu_host_alloc / mgpu_host_register; MGPU_E_INVAL
 * from the feed call otherwise): the host has its list, identical bytes, and builds nothing; its builder stage keeps the two
 * order-dependent sums.  mgpu_collect() then waits for the feed's last k_build_messages and returns the count (no copy when
 * `out` is that array).  The mode for hosts that take struct modesMessage fields on the CPU from an HBM-resident pipeline. */
int mgpu_set_device_messages(mgpu_ctx *ctx, int on);
/* Mode 1: the NEXT feed's records are built into the caller's own device buffer (capacity records) instead of the library's list —
 * e.g. the fixed-size buffer an RCCL gather sends from (readsb_amd/gather.py: no device-to-device copy between the demodulator
 * and the collective; the buffer's lifetime is the caller's).  mgpu_collect_device() then returns that pointer.  NULL: back to the
 * library's list.  Named before every feed that wants it, like mgpu_set_message_buffer. */
int mgpu_set_device_message_buffer(mgpu_ctx *ctx, struct mgpu_msg *d_buf, uint64_t capacity);
int mgpu_collect_device(mgpu_ctx *ctx, const struct mgpu_msg **d_msgs, uint64_t *n, struct mgpu_counters *counters);

/* Same, for IQ already resident in device memory (HBM): d_iq is a device pointer to
 * nsamples samples of cfg.format.  This is the entry the benchmark times. */
int mgpu_feed_iq_device(mgpu_ctx *ctx, const void *d_iq, uint64_t nsamples);

/* Host -> HBM copy only (no processing): stages nsamples IQ samples in the context's device
 * input buffer, whose address mgpu_device_iq_buffer() returns; follow with
 * mgpu_feed_iq_device(ctx, mgpu_device_iq_buffer(ctx), nsamples) to demodulate them in place. */
int   mgpu_upload_iq(mgpu_ctx *ctx, const void *iq_host, uint64_t nsamples);
void *mgpu_device_iq_buffer(mgpu_ctx *ctx);

/* The CPUs the context pinned its host threads to (one physical core each, one L3; 0 = not pinned, e.g. with
 * MGPU_NO_AFFINITY=1).  An application that wants the full speed keeps its own busy threads off these cores and
 * their SMT siblings: a thread of the application sharing a core with a pipeline stage costs up to 30 %.
 * The cores are chosen among the CPUs the process could run on when the library was LOADED (cgroups, taskset) — not among those
 * the creating thread may use at the moment, so that an application following the advice above can create further contexts. */
int mgpu_host_cpus(mgpu_ctx *ctx, int32_t *cpus, int32_t cap);

/* Page-locked host memory placed on the device's NUMA node (hipHostMalloc while the calling thread sits on that node), for the
 * arrays the library writes at full speed: the message arrays of mgpu_set_message_buffer (an aggregator's staging buffers that
 * go on to the GPU), sample buffers.  Pinned memory allocated from a thread on the other socket costs the builder ~10 % of the
 * step.  Free with mgpu_host_free. */
void *mgpu_host_alloc(mgpu_ctx *ctx, uint64_t bytes);
void  mgpu_host_free(mgpu_ctx *ctx, void *ptr);

/* Optional: page-lock a host buffer the caller keeps feeding from (the SDR plugin's read buffer, the
 * ifile reader's `readbuf`, sdr_ifile.c:140) so that mgpu_feed_iq's chunked uploads run at PCIe speed
 * and overlap the kernels.  Unregister before freeing the buffer. */
int mgpu_host_register(mgpu_ctx *ctx, void *ptr, uint64_t bytes);
int mgpu_host_unregister(mgpu_ctx *ctx, void *ptr);

/* EOF handling of ifileRun: when the stream length is an exact multiple of buf_samples the
 * reference pushes one more zero-length buffer (sdr_ifile.c:223-237).  Call once at end. */
int mgpu_finish(mgpu_ctx *ctx);

/* Messages of the calls since the last collect, in stream order (netUseMessage order,
 * demod_2400.c:471).  Copies up to `cap` messages, *n = number copied; remaining count
 * via mgpu_pending_messages().  counters may be NULL. */
int mgpu_collect(mgpu_ctx *ctx, struct mgpu_msg *out, uint64_t cap, uint64_t *n,
                 struct mgpu_counters *counters);
/* Optional: have the messages built straight into the caller's array (capacity records) instead of an
 * internal list.  After a feed, mgpu_collect(ctx, buf, capacity, &n, ...) with the same pointer only
 * reports n and makes the array writable again — no copy (the builder threads have written the 64-byte
 * records with streaming stores; 16-byte alignment of buf keeps that fast).  A feed that needs more
 * room than capacity fails with MGPU_E_OVERFLOW.  buf = NULL returns to the internal list.  The array
 * may be replaced between feeds (e.g. alternating staging buffers) once the pending messages are
 * collected. */
int mgpu_set_message_buffer(mgpu_ctx *ctx, struct mgpu_msg *buf, uint64_t capacity);

uint64_t mgpu_pending_messages(mgpu_ctx *ctx);

/* icaoFilterExpire() (icao_filter.c:96-110) / icaoFilterAdd() (:112-130) on the context's filter, between two feed
 * calls.  mgpu_filter_expire is how a host drives the filter with cfg.filter_clock = MGPU_FILTER_CLOCK_EXTERNAL (it is
 * refused with MGPU_E_INVAL in the other modes: two clocks would double-flip); mgpu_filter_add mirrors adds the host
 * makes outside the demodulator (decodeModesMessage on network input, mode_s.c:766-779) and is allowed in every mode. */
int mgpu_filter_expire(mgpu_ctx *ctx);
int mgpu_filter_add(mgpu_ctx *ctx, uint32_t addr);

int mgpu_last_timing(mgpu_ctx *ctx, struct mgpu_timing *t);
/* What mgpu_timing's per-kernel figures (convert_ms, sweep_ms, slice_ms: a pair of timing HIP events around ONE kernel in a busy
 * stream) contain beyond the kernel's own duration: measured here, now, with a kernel of known duration (it spins 20 us on the
 * GPU's 100 MHz counter) between two such events — 4.0 us on an MI355X under ROCm 7.2, whatever the kernel's length
 * (tools/micro/event_overhead.hip).  bench.py subtracts it from its per-launch figures and reports both. */
int mgpu_event_bracket_us(mgpu_ctx *ctx, float *us);

/* ---- the two plugin-surface pieces on their own ----------------------------------- */

/* iq_convert_fn (convert.h:34-39) for the non-DC-filter converters convert_uc8_nodc,
 * convert_sc16_nodc, convert_sc16q11_nodc (convert.c:64,212,329): nsamples IQ samples in
 * host memory -> nsamples u16 magnitudes in host memory; out_mean_* may be NULL.  Not in the middle of a stream fed through
 * mgpu_feed_iq* on the same context (MGPU_E_INVAL: it would overwrite the stream's 326-sample tail): one context per role. */
int mgpu_convert(mgpu_ctx *ctx, const void *iq_host, uint16_t *mag_host, uint32_t nsamples,
                 double *out_mean_level, double *out_mean_power);

/* demodulate2400(struct mag_buf *) (demod_2400.h:38) on one magnitude buffer laid out as
 * struct mag_buf.data (readsb.h:450-464): trailing_samples of overlap then `length` new
 * samples.  The caller passes the struct's scalar fields; messages come back through
 * mgpu_collect().  The stream position / filter clock advance exactly as for mgpu_feed_iq.
 * dropped: nonzero = the host has recently dropped samples (what demod_2400.c:335-338 reads from
 * Modes.stats_15min.samples_dropped): this buffer is swept with max(PREAMBLE_THRESHOLD_PIZERO = 75,
 * cfg.preamble_threshold), as the reference does. */
int mgpu_demod_mag_buf(mgpu_ctx *ctx, const uint16_t *data, uint32_t length,
                       int64_t sampleTimestamp, int64_t sysTimestamp,
                       double mean_power, uint32_t dropped);
/* The same when cfg.mode_ac is set — demodulate2400(buf) followed by demodulate2400AC(buf) (readsb.c:871-874): the Mode A/C
 * demodulator derives its noise floor from mag_buf.mean_level and mean_power (demod_2400.c:579-580), so the caller passes
 * both.  (mgpu_demod_mag_buf on a mode_ac context is refused with MGPU_E_INVAL rather than decoded without Mode A/C.) */
int mgpu_demod_mag_buf_ac(mgpu_ctx *ctx, const uint16_t *data, uint32_t length,
                          int64_t sampleTimestamp, int64_t sysTimestamp,
                          double mean_level, double mean_power, uint32_t dropped);

/* ---- one capture sharded by buffer ranges over several contexts / GPUs (BASELINE config 5) ----
 * Buffers are independent except for the ICAO filter; the pre-screen needs (a superset of) the addresses the filter may
 * hold: those some clean DF17 / DF11-IID0 frame carried within the last two filter generations (120 s of samples).
 *   the context that owns the capture's FIRST range: mgpu_reset; mgpu_feed_iq*(its samples) as for any stream;
 *   every other range (a context, after mgpu_reset):
 *     pass 1: mgpu_shard_begin(ctx, range_first - 120 s (whole buffers, >= 0), history, 1); mgpu_feed_iq*(those samples up to
 *             range_first); mgpu_adder_bitmap_get()                       -> the addresses that may still be known at range_first
 *     pass 2: mgpu_reset; mgpu_adder_bitmap_set(that); mgpu_shard_begin(ctx, range_first, history, 2);
 *             mgpu_feed_iq*(the range's samples); mgpu_shard_packets()    -> the range's live records
 *   the first context again: mgpu_walk_packets() over the other ranges' packets in stream order, mgpu_finish(),
 *   mgpu_collect(): the message list AND every counter of the unsharded stream, bit for bit.
 * (Any superset of the needed addresses is exact too, e.g. the OR of pass-1 bitmaps over all ranges.)
 * A packet (one per chunk of the shard's range, 8-byte aligned) carries what the walking rank cannot compute without
 * the samples: per live record its would-be signal power and the counts of its would-be skip window, per buffer the
 * converter's level / power sums, per chunk the sweep's candidate tallies.
 * first_sample is a multiple of buf_samples; history = the 326 IQ samples before it (NULL for 0). */
int mgpu_shard_begin(mgpu_ctx *ctx, uint64_t first_sample, const void *history_iq, int mode);
int mgpu_adder_bitmap_get(mgpu_ctx *ctx, uint32_t *words /* 2^19 */);
int mgpu_adder_bitmap_set(mgpu_ctx *ctx, const uint32_t *words /* 2^19 */);
int mgpu_shard_packets(mgpu_ctx *ctx, const void **packets, uint64_t *bytes);   /* valid until the next reset */
int mgpu_walk_packets(mgpu_ctx *ctx, const void *packets, uint64_t bytes);   /* continues the context's stream: the packets' first sample = where it stands */

/* ---- config 5 with the ordered walk itself sharded across the ranks (round 4) ----
 * The form above leaves the ordered walk and the message build of the WHOLE capture to the first rank; this one has every
 * rank walk and build its own range.  What ties the ranges together is the ICAO filter (icao_filter.c:65-130: two generations,
 * `occupied`, the table size with its grow / shrink hysteresis) and the clock of its expiry, which is data-dependent: after a
 * buffer the reference tests Modes.synthetic_now — the timestamp of the buffer's last scored candidate, demod_2400.c:412-414 —
 * against next_flip and re-arms it 60 s after THAT (readsb.c:1227-1231).  Protocol (readsb_amd/shard.py: ShardWalkRank):
 *   1. rank r: mgpu_reset; mgpu_shard_begin(ctx, warmup_first, history, 2); mgpu_feed_iq*(warm-up: the two filter generations
 *      before its range), mgpu_feed_iq*(its range) — two feeds, so that no packet straddles the range's first sample;
 *   2. mgpu_shard_clock_estimate(): its buffers' end clocks, estimated from the records alone; all-gather;
 *      mgpu_flip_schedule() over all buffers' clocks = the expiry schedule;
 *   3. mgpu_shard_walk(): warm-up + range walked with that schedule imposed (rank 0, whose packets start at sample 0: from the
 *      reference's initial state; the others: from an empty filter at the warm-up's first sample), the range's messages built,
 *      its counters kept; out: the range's TRUE end clocks, the filter state at the range's first sample and at its end
 *      (mgpu_shard_state);
 *   4. all-gather clocks and states.  Done iff mgpu_flip_schedule over the true clocks reproduces the schedule and every rank's
 *      state at its first sample equals the state the rank before it ended with (byte-equal blobs).  Otherwise again from 3 with
 *      the new schedule, a rank whose seam failed starting from its predecessor's end state (args.start_state);
 *   5. the last rank: mgpu_finish (the EOF buffer; its clock belongs to the schedule too); every rank: mgpu_collect — its range's
 *      messages and counters.  Integer counters add up over the ranks; nflips is the last rank's; peak_signal_power the maximum;
 *      signal_power_sum / noise_power_sum are sequential double sums in the reference (demod_2400.c:445-479): re-add them in
 *      stream order with mgpu_seqsum_signal_power (over the gathered messages) / mgpu_seqsum (over mgpu_shard_noise_terms).
 * At the fixed point every rank's walk is the serial walk's.  tests/test_gpu_shard.py, tests/test_shard_walk.py. */
struct mgpu_shard_walk_args {
    uint64_t own_first;          /* first sample of the rank's own range (a multiple of buf_samples); packets before it are warm-up */
    const int64_t *flip_after;   /* the imposed expiry schedule: sampleTimestamp (12 MHz ticks = sample * 5) of every buffer of the
                                  * CAPTURE after which the filter expires, ascending */
    uint64_t nflips;
    const void *start_state;     /* NULL: start at the first packet (see 3. above); else the filter state at own_first — the end
                                  * state of the rank before — and the warm-up packets are skipped */
    uint64_t start_state_bytes;
    int32_t check_records;       /* 1: validate every record of the packets (packets that came over a network) */
    int32_t reserved;
};
/* packets == NULL: the context's own (mgpu_shard_packets).  end_clocks[cap]: the range's buffers' end clocks (ms), *n of them. */
int mgpu_shard_clock_estimate(mgpu_ctx *ctx, const void *packets, uint64_t bytes, uint64_t own_first,
                              int64_t *end_clocks, uint64_t cap, uint64_t *n);
int mgpu_shard_walk(mgpu_ctx *ctx, const void *packets, uint64_t bytes, const struct mgpu_shard_walk_args *args,
                    int64_t *end_clocks, uint64_t cap, uint64_t *n);
int mgpu_shard_state(mgpu_ctx *ctx, int which /* 0: at own_first, 1: at the range's end */, const void **blob, uint64_t *bytes);
int mgpu_shard_noise_terms(mgpu_ctx *ctx, const double **terms, uint64_t *n);   /* per buffer of the range: its addend to noise_power_sum */
int mgpu_shard_signal_terms(mgpu_ctx *ctx, const uint64_t **terms, uint64_t *n);   /* per accepted message of the range: sig_sumsq (n = 0 with Mode A/C: use the messages) */
/* The same rank with its pass through the ORDINARY pipeline — ordered walk and message build overlapped with the kernels, as for any
 * stream — for when the schedule is known before the pass (shard.py derives it from a pre-pass over the few buffers around every
 * expiry's possible positions):  mgpu_shard_stream_begin (resets the context; first_sample = the warm-up's first sample, or own_first
 * with start_state) | mgpu_feed_iq*(warm-up) | mgpu_shard_stream_mark (the state at own_first is kept, clocks are logged from
 * there; the warm-up leaves no messages and no statistics) | mgpu_feed_iq*(range) with mgpu_collect as for any stream |
 * mgpu_shard_stream_end -> the range's true end clocks; states by mgpu_shard_state, noise terms by mgpu_shard_noise_terms.
 * Synchronous or DEFERRED feeds (mgpu_set_deferred before _begin): with deferred feeds nothing waits at the mark — the walker marks
 * the range's begin itself when it reaches the range's first chunk, and the range's kernels run while the warm-up is still being
 * walked (one pipeline fill and drain per pass instead of two); every feed call still has its own message list to be collected,
 * the warm-up's are empty.
 * The rounds that follow are those of mgpu_shard_walk; a rank whose premise failed runs its pass again. */
struct mgpu_shard_stream_args {
    uint64_t first_sample;       /* where the pass starts (a multiple of buf_samples) */
    const void *history_iq;      /* the 326 IQ samples before it (NULL for 0) */
    uint64_t own_first;          /* first sample of the rank's own range */
    const int64_t *flip_after;   /* the imposed expiry schedule (as in mgpu_shard_walk_args) */
    uint64_t nflips;
    const void *start_state;     /* NULL, or the filter state at own_first (then first_sample == own_first) */
    uint64_t start_state_bytes;
};
int mgpu_shard_stream_begin(mgpu_ctx *ctx, const struct mgpu_shard_stream_args *args);
int mgpu_shard_stream_mark(mgpu_ctx *ctx);
int mgpu_shard_stream_end(mgpu_ctx *ctx, int64_t *end_clocks, uint64_t cap, uint64_t *n);
/* The expiry schedule from every buffer's end clock, by the reference's rule (readsb.c:1227-1231; filter_clock as in
 * mgpu_config): returns the number of expiries, flip_after[i < cap] = index of the buffer the i-th one follows. */
uint64_t mgpu_flip_schedule(const int64_t *end_clock, uint64_t nbuf, int64_t startup_ms, int filter_clock,
                            uint64_t *flip_after, uint64_t cap);
/* The buffers an expiry of the filter CAN follow, whatever the data (mask[b] = 1; returns how many): what the stream form's pre-pass
 * looks at.  T_k + 60 000 <= T_(k+1) < T_k + 60 111 ms for the clocks T_k the expiries are due at, so the windows widen by 111 ms per
 * expiry: ~6 % of a one-hour capture's buffers. */
uint64_t mgpu_expiry_windows(uint64_t nbuf_total, uint32_t buf_samples, int64_t startup_ms, int filter_clock, uint8_t *mask);
/* What every rank concludes from a round's all-gather (step 4 above; the same on every rank): the schedule the true clocks give
 * (next_sched[cap], *n_next entries), *done = it is the imposed one AND every range's state at its first sample equals the state the
 * range before it ended with; import_from[r] = the rank whose end state rank r has to start its next pass from, or -1.  clocks[r] /
 * nclocks[r]: rank r's end clocks (an empty range: 0 of them; its states are ignored). */
int mgpu_shard_round(const int64_t *sched, uint64_t nsched, uint32_t world, const int64_t *const *clocks, const uint64_t *nclocks,
                     const void *const *state_first, const uint64_t *state_first_bytes, const void *const *state_end, const uint64_t *state_end_bytes,
                     uint64_t nsamples, uint32_t buf_samples, int64_t startup_ms, int filter_clock,
                     int64_t *next_sched, uint64_t cap, uint64_t *n_next, int32_t *import_from, int32_t *done);
double mgpu_seqsum(double start, const double *terms, uint64_t n);                               /* ((start + t0) + t1) + ... */
double mgpu_seqsum_signal_power(double start, const struct mgpu_msg *msgs, uint64_t n);          /* ... of sig_sumsq / 65535^2 */
/* The same sequential sum of the messages' signal powers, prepared by ranges and applied in O(blocks) (readsb_amd/csrc/seqsum.cpp):
 * a block of `block` messages is ONE integer addition on the running sum's mantissa while the sum stays in the binade the block
 * was prepared for (every addition then rounds to the same grid; ties carry the parity of the steps since the tie before).
 * mgpu_seqsum_blocks (every rank, over its own messages): approx_start = roughly what the sum is where the range begins (the
 * earlier ranges' totals, added any which way); out[ceil(n / block)].  mgpu_seqsum_apply (the combining rank, ranges in stream
 * order, start = the exact sum so far): blocks whose premise fails — the sum is not in the predicted binade, or leaves it — are
 * re-added message by message (*fallbacks counts them; may be NULL).  Returns exactly mgpu_seqsum_signal_power(start, msgs, n). */
struct mgpu_sum_block { uint64_t total; int32_t e; uint32_t flags; };
int    mgpu_seqsum_blocks(double approx_start, const struct mgpu_msg *msgs, uint64_t n, uint32_t block, struct mgpu_sum_block *out);
/* the same blocks from the 8-byte numerators of the messages' signal powers (mgpu_shard_signal_terms: one per accepted message of the
 * range, logged by the builder during the pass; none with Mode A/C) instead of from the 64-byte messages — an eighth of the memory read */
int    mgpu_seqsum_blocks_terms(double approx_start, const uint64_t *sumsq, uint64_t n, uint32_t block, struct mgpu_sum_block *out);
double mgpu_seqsum_apply(double start, const struct mgpu_msg *msgs, uint64_t n, uint32_t block, const struct mgpu_sum_block *blocks,
                         uint64_t *fallbacks);

/* ---- beast wire output (modesSendBeastOutput, net_io.c:1655-1714) ------------------------------------
 * Per message: 0x1a, type '2' (56-bit) / '3' (112-bit) / '1' (Mode A/C), the 12 MHz timestamp as 6 bytes
 * big-endian, one signal byte clamp(nearbyint(sqrt(signalLevel) * 255), 1..255), the (corrected) message
 * bytes; every 0x1a payload byte doubled; no 0x1a 0xe3 receiverId prefix.  Encoded on the GPU.
 * _device: d_msgs / d_out are device pointers (e.g. the records an aggregator gathered into its HBM);
 * *bytes = size of the stream; MGPU_E_OVERFLOW (with *bytes set) if it does not fit cap. */
int mgpu_beast_encode_device(mgpu_ctx *ctx, const struct mgpu_msg *d_msgs, uint64_t n, uint8_t *d_out, uint64_t cap, uint64_t *bytes);
int mgpu_beast_encode(mgpu_ctx *ctx, const struct mgpu_msg *msgs, uint64_t n, uint8_t *out, uint64_t cap, uint64_t *bytes);

/* ---- per-message field decode (decodeModesMessage behind the CRC stage, mode_s.c:598-760; decodeExtendedSquitter and
 * its ME decoders, mode_s.c:806-1555; decodeCommB, comm_b.c; decodeModeAMessage, mode_ac.c:171-200) ---------------------------------------
 * One record per message, same index, decoded on the GPU from the corrected frame: the raw Annex-10 fields, altitude,
 * squawk, callsign, velocity, CPR words, the accuracy / operational-status / target-state groups — the members of
 * struct modesMessage (readsb.h:887-1143) under the same names; enums carry the reference's numeric values
 * (addrtype_t readsb.h:178, datasource_t :159, airground_t :216, heading_type_t :239, sil_type_t :224, cpr_type_t :229,
 * nav_modes_t :263, nav_altitude_source_t :287, emergency_t :275, altitude_unit_t :204).  What a memset-0 modesMessage
 * would hold stays 0.  MB/MD/ME/MV are msg[4..10] / msg[1..10] of the message record and are not repeated.
 * DF20/21: decodeCommB's register inference (comm_b.c:52-961) fills commb_format (commb_format_t, readsb.h:249) and the
 * fields of the register it settles on (callsign; selected altitude / QNH / modes; roll, track, ground speed, track
 * rate, TAS; heading, IAS, Mach, vertical rates; the BDS4,4 weather group).  mach is the float the reference stores
 * into its double.  176 bytes. */
struct mgpu_fields {
    uint32_t addr;              /* mm->addr (with MODES_NON_ICAO_ADDRESS = 1<<24 where the ME decode says so) */
    uint32_t AA;
    uint32_t flags;             /* MGPU_F_* */
    uint16_t acc_flags;         /* MGPU_ACC_* */
    uint8_t  nav_flags;         /* MGPU_NAV_* */
    uint8_t  msgtype;           /* DF, 77 = Mode A/C */
    uint8_t  addrtype, source, airground, metype;
    uint8_t  mesub, CA, CC, CF;
    uint8_t  DR, FS, KE, ND;
    uint8_t  RI, SL, UM, VS;
    uint8_t  IID, category, emergency, cpr_type;
    uint16_t AC, ID;
    uint16_t squawkHex, squawkDec;
    int32_t  baro_alt, geom_alt;
    int32_t  geom_delta, baro_rate, geom_rate;
    uint16_t ias, tas;
    float    heading, gs_v0, gs_v2, gs_selected;
    uint32_t cpr_lat, cpr_lon;
    char     callsign[8];
    uint8_t  baro_alt_unit, geom_alt_unit, heading_type, sil_type;
    uint8_t  nac_p, nac_v, sil, gva;
    uint8_t  sda, op_version, op_hrd, op_tah;
    uint16_t op_flags;          /* MGPU_OP_* */
    uint8_t  op_cc_lw, op_cc_antenna_offset;
    uint8_t  op_cc_tc, nav_heading_type, nav_altitude_source, nav_modes;
    uint32_t nav_fms_altitude, nav_mcp_altitude;
    float    nav_qnh, nav_heading;
    float    roll, track_rate, mach;          /* Comm-B BDS5,0 / 6,0 (mm->mach is a double holding this float) */
    float    oat, humidity, wind_direction;   /* Comm-B BDS4,4 */
    uint16_t wind_speed, static_pressure;
    uint8_t  commb_format, met_source, turbulence, pad0;
    uint8_t  reserved[8];
};

/* flags: the bools of struct modesMessage (readsb.h:954-993) */
#define MGPU_F_BARO_ALT_VALID   (1u << 0)
#define MGPU_F_GEOM_ALT_VALID   (1u << 1)
#define MGPU_F_HEADING_VALID    (1u << 2)
#define MGPU_F_GS_VALID         (1u << 3)
#define MGPU_F_IAS_VALID        (1u << 4)
#define MGPU_F_TAS_VALID        (1u << 5)
#define MGPU_F_BARO_RATE_VALID  (1u << 6)
#define MGPU_F_GEOM_RATE_VALID  (1u << 7)
#define MGPU_F_SQUAWK_VALID     (1u << 8)
#define MGPU_F_CALLSIGN_VALID   (1u << 9)
#define MGPU_F_CPR_VALID        (1u << 10)
#define MGPU_F_CPR_ODD          (1u << 11)
#define MGPU_F_CATEGORY_VALID   (1u << 12)
#define MGPU_F_GEOM_DELTA_VALID (1u << 13)
#define MGPU_F_SPI_VALID        (1u << 14)
#define MGPU_F_SPI              (1u << 15)
#define MGPU_F_ALERT_VALID      (1u << 16)
#define MGPU_F_ALERT            (1u << 17)
#define MGPU_F_EMERGENCY_VALID  (1u << 18)
#define MGPU_F_ALT_Q_BIT        (1u << 19)
#define MGPU_F_ACAS_RA_VALID    (1u << 20)
#define MGPU_F_ROLL_VALID       (1u << 21)
#define MGPU_F_TRACK_RATE_VALID (1u << 22)
#define MGPU_F_MACH_VALID       (1u << 23)
#define MGPU_F_WIND_VALID       (1u << 24)
#define MGPU_F_OAT_VALID        (1u << 25)
#define MGPU_F_STATIC_PRESSURE_VALID (1u << 26)
#define MGPU_F_TURBULENCE_VALID (1u << 27)
#define MGPU_F_HUMIDITY_VALID   (1u << 28)
#define MGPU_F_MET_SOURCE_VALID (1u << 29)
/* acc_flags: mm->accuracy (readsb.h:1061-1086) */
#define MGPU_ACC_NIC_A_VALID    (1u << 0)
#define MGPU_ACC_NIC_B_VALID    (1u << 1)
#define MGPU_ACC_NIC_C_VALID    (1u << 2)
#define MGPU_ACC_NIC_BARO_VALID (1u << 3)
#define MGPU_ACC_NAC_P_VALID    (1u << 4)
#define MGPU_ACC_NAC_V_VALID    (1u << 5)
#define MGPU_ACC_GVA_VALID      (1u << 6)
#define MGPU_ACC_SDA_VALID      (1u << 7)
#define MGPU_ACC_NIC_A          (1u << 8)
#define MGPU_ACC_NIC_B          (1u << 9)
#define MGPU_ACC_NIC_C          (1u << 10)
#define MGPU_ACC_NIC_BARO       (1u << 11)
/* nav_flags: mm->nav (readsb.h:1127-1142) */
#define MGPU_NAV_HEADING_VALID  (1u << 0)
#define MGPU_NAV_FMS_ALT_VALID  (1u << 1)
#define MGPU_NAV_MCP_ALT_VALID  (1u << 2)
#define MGPU_NAV_QNH_VALID      (1u << 3)
#define MGPU_NAV_MODES_VALID    (1u << 4)
/* op_flags: the one-bit members of mm->opstatus (readsb.h:1103-1120) */
#define MGPU_OP_VALID       (1u << 0)
#define MGPU_OP_OM_ACAS_RA  (1u << 1)
#define MGPU_OP_OM_IDENT    (1u << 2)
#define MGPU_OP_OM_ATC      (1u << 3)
#define MGPU_OP_OM_SAF      (1u << 4)
#define MGPU_OP_CC_ACAS     (1u << 5)
#define MGPU_OP_CC_CDTI     (1u << 6)
#define MGPU_OP_CC_1090_IN  (1u << 7)
#define MGPU_OP_CC_ARV      (1u << 8)
#define MGPU_OP_CC_TS       (1u << 9)
#define MGPU_OP_CC_UAT_IN   (1u << 10)
#define MGPU_OP_CC_POA      (1u << 11)
#define MGPU_OP_CC_B2_LOW   (1u << 12)
#define MGPU_OP_CC_LW_VALID (1u << 13)

/* msgs / out in host memory (copied through the context's device scratch), or both in device memory. */
int mgpu_decode_fields(mgpu_ctx *ctx, const struct mgpu_msg *msgs, uint64_t n, struct mgpu_fields *out);
int mgpu_decode_fields_device(mgpu_ctx *ctx, const struct mgpu_msg *d_msgs, uint64_t n, struct mgpu_fields *d_out);

/* ---- first stage of the tracker + the forwarding rule --------------------------------------------------------
 * What the reference decides about an accepted message before and around its position tracker, on the device, over the message
 * list: address_reliable (track.c:1688-1693); whether trackUpdateFromMessage finds or creates the message's aircraft and counts it
 * (track.c:1905-1966: aircraftGet / aircraftCreate for reliable addresses only, a->seen, the 45 s rule for the formats whose
 * address is the CRC residue, a->messages++); and outputMessage's rule (net_io.c:5846-5849) over the batch drainMessageBuffer
 * hands it (net_io.c:5924-5940: one sample buffer's messages, cut every 256, all updates before all outputs):
 *     forwarded  <=>  (crc == 0 && correctedbits == 0)  ||  (mm->aircraft && mm->aircraft->messages > 1)  ||  Mode A/C
 * (the beast and raw outputs additionally want correctedbits < 2, net_io.c:5863-5872: the caller's test, it needs nothing from here).
 * The position tracker itself (cpr.c, the speed / range checks, track.c:423-745) stays on the host: when it judges a position
 * message bad or duplicate it puts the aircraft's copy back, a->messages and a->seen with it (track.c:2625-2627), and it deletes
 * aircraft without a reliable position after 5 silent minutes (track.c:2856-2880).  The gate carries both as BOUNDS per aircraft
 * and answers "deferred" exactly where the bounds disagree — in practice inside an aircraft's first two messages only: 0.2 % of a
 * 200-aircraft minute (tests/golden/gate_*.npz: every other verdict equals what the whole reference program forwarded).
 *   verdict[i]: bits 0-1  MGPU_GATE_DROP / _FORWARD / _DEFER;  bit 2  address_reliable;  bit 3  mm->aircraft may be set;
 *               bit 4  mm->aircraft is set for certain.
 * msgs: the accepted messages in netUseMessage order (what mgpu_collect returns), WHOLE sample buffers per call, timestamps on the
 * ifile grid (buffer = (timestamp - 772) / 5 / cfg.buf_samples).  The aircraft table (device memory, 1.25 GiB, allocated by the first
 * call) lives from call to call until mgpu_track_gate_reset / mgpu_destroy.
 * Long streams: removeStaleRange also deletes an aircraft WITH a reliable position once that position is an hour old (30 minutes:
 * non-ICAO addresses; track.c:2835-2866).  Which position was the last reliable one is the tracker's knowledge, so from an aircraft's
 * first position message + that timeout on, its non-clean messages (repaired bits, Address/Parity formats) are answered DEFER: certain
 * verdicts stay what the reference did however long the stream runs, and an aircraft heard for more than an hour costs the host's
 * tracker its non-clean messages.
 * These entries (and the field decode / beast encoder) run on a stream of their own: they do not wait for chunks a deferred feed has
 * queued.  One context is not thread-safe, and the entries share its staging buffers. */
#define MGPU_GATE_DROP     0
#define MGPU_GATE_FORWARD  1
#define MGPU_GATE_DEFER    2
#define MGPU_GATE_ADDRESS_RELIABLE  4
#define MGPU_GATE_AIRCRAFT_POSSIBLE 8
#define MGPU_GATE_AIRCRAFT_CERTAIN  16
int mgpu_track_gate(mgpu_ctx *ctx, const struct mgpu_msg *msgs, uint64_t n, uint8_t *verdict);                 /* host arrays; decodes the fields it needs itself */
int mgpu_track_gate_device(mgpu_ctx *ctx, const struct mgpu_msg *d_msgs, const struct mgpu_fields *d_fields, uint64_t n, uint8_t *d_verdict);   /* everything in device memory */
int mgpu_track_gate_reset(mgpu_ctx *ctx);

/* The gate applied to the encoder — what an aggregator that only forwards needs of the tracker: the beast stream
 * of the messages the reference forwards FOR CERTAIN (outputMessage, net_io.c:5822-5885: first messages of an aircraft are suppressed
 * unless crc == 0 && correctedbits == 0, :5846-5849), in stream order, and — in stream order too — the list of the messages whose fate
 * is the position tracker's (MGPU_GATE_DEFER): {index in msgs, the offset in the stream its frame would start at}.  The host's tracker
 * settles those few (0.2 % of a 200-aircraft minute), encodes the ones it forwards (mgpu_beast_encode on that handful) and splices them
 * in at their offsets: the result is byte for byte what the whole reference program writes (tests/test_gpu_beast.py against
 * tests/golden/beast_*.bin, written by the program's --dump-beast).
 * flags: MGPU_BEAST_NET_RULE = also the beast / raw NETWORK outputs' test correctedbits < 2 (net_io.c:5863-5872; --net-verbatim
 * lifts it); 0 = the --dump-beast file's rule (no such test).
 * mgpu_beast_encode_gated: msgs in host memory, WHOLE sample buffers per call in netUseMessage order like mgpu_track_gate (it runs the
 * field decode and the gate itself, continuing the context's aircraft table).  _device: messages, verdicts (mgpu_track_gate_device's),
 * stream and list all in device memory.  MGPU_E_OVERFLOW: the stream (*bytes = what it needs) or the list (*ndeferred) does not fit.
 * readsb_amd/host/readsb_gpu_gather.c --forward-only uses it on the gathered records (/root/reference/net_io.c:1655-1714 is what
 * consumes such a stream on the other side). */
struct mgpu_deferred {
    uint64_t index;               /* of the message in msgs */
    uint64_t offset;              /* bytes of certain frames before it: where its frame goes if the tracker forwards it */
};
#define MGPU_BEAST_NET_RULE 1u
int mgpu_beast_encode_gated(mgpu_ctx *ctx, const struct mgpu_msg *msgs, uint64_t n, uint32_t flags, uint8_t *out, uint64_t cap, uint64_t *bytes,
                            struct mgpu_deferred *deferred, uint64_t deferred_cap, uint64_t *ndeferred);
int mgpu_beast_encode_gated_device(mgpu_ctx *ctx, const struct mgpu_msg *d_msgs, const uint8_t *d_verdict, uint64_t n, uint32_t flags,
                                   uint8_t *d_out, uint64_t cap, uint64_t *bytes, struct mgpu_deferred *d_deferred, uint64_t deferred_cap,
                                   uint64_t *ndeferred);

/* ---- tables, for known-answer tests against crc.c --------------------------------- */

/* These run on the host (they are how the device tables are built) and need no context. */
uint32_t mgpu_crc_checksum(const uint8_t *msg, int bits);                 /* modesChecksum, crc.c:67 */
/* modesChecksumDiagnose (crc.c:383) for the tables modesChecksumInit(nfix_crc) builds:
 * returns #error bits 0..2 (positions in *bit0,*bit1, -1 = unused), or -1 if uncorrectable */
int mgpu_crc_diagnose(int nfix_crc, uint32_t syndrome, int bits, int *bit0, int *bit1);
int mgpu_crc_table_size(int nfix_crc, int bits);                           /* crctests' table sizes */
/* the 65536-entry UC8 magnitude table (init_uc8_lookup, convert.c:35-62), index I | Q<<8 */
const uint16_t *mgpu_uc8_table(void);

/* ---- diagnostics ------------------------------------------------------------------------------
 * Host-logic self-check, needs no GPU: walks a seeded synthetic record stream (aircraft that appear,
 * go quiet long enough to expire from the ICAO filter, and return) once serially and once as
 * `nsegments` speculative buffer ranges per chunk, and compares every decision, counter and the
 * final filter.  0 = identical, k > 0 = first differing chunk + 1, < 0 = bad arguments.
 * *speculated_permille (may be NULL) = share of chunks whose ranges all committed in the first batch. */
int mgpu_selftest_walk(uint64_t seed, uint32_t nchunks, uint32_t buffers_per_chunk, uint32_t nsegments,
                       uint32_t naircraft, uint32_t *speculated_permille);

/* Which of a node's GPUs a PCI function is — its place among the functions with the same vendor, device id and local CPU list
 * under `pci_devices_dir` (the library reads /sys/bus/pci/devices), by bus address; what the pipeline's threads choose their L3
 * groups by (a container that holds ONE of a node's GPUs sees HIP ordinal 0 whichever it is).  -1 = not found, -2 = bad arguments. */
int mgpu_selftest_device_index(const char *pci_devices_dir, const char *bus_id);

/* Same idea for the walk on the device (below), needs no GPU either: its algorithm restated on the host — every buffer walked on
 * its own against the filter at the start of the chunk plus a table of first adds, iterated (at most max_walks times) until the
 * table reproduces itself, then the premise check — against the serial walk on the same seeded streams.  Chunks that do not
 * settle or whose premises fail are walked serially, as the library does.  0 = identical, k > 0 = first differing chunk + 1.
 * stats (may be NULL): [0] chunks decided by the model, [1] chunks walked serially after all, [2] walks in all, [3] most walks
 * one chunk took. */
int mgpu_selftest_device_walk(uint64_t seed, uint32_t nchunks, uint32_t buffers_per_chunk, uint32_t naircraft,
                              uint32_t max_walks, uint64_t stats[4]);

/* The sharded walk's protocol (mgpu_shard_walk above), needs no GPU: a seeded synthetic record stream of nchunks chunks is dealt
 * to `nranks` ranks in whole chunks; every rank walks warm-up + range (nsegments speculative buffer ranges per chunk, 1 = the
 * serial loop) from an empty filter with the schedule from ESTIMATED end clocks imposed, and the rounds over (schedule, seam
 * states) run as they would with all-gathers in between — against the serial walk of the whole stream: every decision of every
 * chunk, every buffer's end clock, the counts, the final filter state.  front_extra more aircraft transmit during the first 150 s
 * only (the filter's table grows for them and shrinks one size per expiry afterwards: state no warm-up can rebuild, so seams fail
 * and ranks import).  flags bit 0: the first schedule from the buffers' start clocks; bit 1: a deliberately wrong first schedule.
 * 0 = identical, k > 0 = first differing chunk + 1, -2 = the rounds did not settle.  stats (may be NULL): [0] rounds, [1] walks
 * of a range in all, [2] seams that failed in some round, [3] rounds in which the schedule changed, [4] expiries, [5] ranges that
 * ended up starting from an imported state. */
int mgpu_selftest_shard_walk(uint64_t seed, uint32_t nchunks, uint32_t buffers_per_chunk, uint32_t naircraft, uint32_t front_extra,
                             uint32_t nranks, uint32_t nsegments, uint32_t flags, uint64_t stats[6]);

/* The ordered walk on the device (environment MGPU_DEVICE_WALK=1, or =check to run it beside the host walk and compare every
 * decision): out[0] chunks, [1] chunks whose decisions came from the device (check: were compared), [2] chunks the fixed point
 * did not settle on in time, [3] chunks whose premises failed afterwards (the filter table grew, the expiry moved), [4] chunks
 * the walk does not model, [5] walks run in all, [6] differences found (check mode; must be 0), [7] reserved.
 * [2]-[4] are walked on the host. */
int mgpu_debug_device_walk(mgpu_ctx *ctx, uint64_t out[8]);

/* The magnitudes the pipeline's LAST chunk was demodulated from, as they lie in HBM: out[i] = magnitude of the chunk's sample i,
 * i < n <= the chunk's length (MGPU_E_INVAL beyond it, or before any feed).  Since round 6 the pipeline's magnitudes are not
 * mgpu_convert()'s kernels' — for UC8 input they come out of k_sweep_uc8 (converter and sweep in one pass), for SC16 / SC16Q11 out
 * of k_sweep_sc16 — so the exhaustive converter tests (every UC8 byte pair, every 12-bit SC16Q11 pair: convert.c:64-108, 212-250,
 * 329-367) are run against this too (tests/test_gpu_convert.py).  Drains the pipeline first.  Test support: nothing in readsb
 * reads magnitudes back. */
int mgpu_debug_last_magnitudes(mgpu_ctx *ctx, uint16_t *out, uint64_t n);

#ifdef __cplusplus
}
#endif
#endif
