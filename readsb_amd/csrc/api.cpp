// api.cpp — C ABI of libmodes_gpu.so (include/modes_gpu.h): context, device memory, the
// per-feed pipeline  convert -> sweep/slice -> pre-screen -> ordered walk -> signal power,
// and the counters the reference keeps in Modes.stats_current.
//
// The product has no CPU compute path: without a usable HIP device mgpu_create() fails with
// MGPU_E_NODEVICE.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdio>
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

namespace {
double wall_ms() {
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}
}  // namespace

struct mgpu_ctx {
    mgpu_config cfg{};
    hipStream_t stream = nullptr;
    hipEvent_t ev[8] = {};
    std::string err;

    // capacities
    uint64_t cap_samples = 0, cap_units = 0, cap_buffers = 0, cap_pool = 0, cap_msgs = 0;

    // device
    uint8_t *d_iq = nullptr;
    uint16_t *d_mag = nullptr, *d_tail = nullptr;
    PhaseRec *d_pool = nullptr, *d_live = nullptr;
    uint32_t *d_pool_used = nullptr, *d_unit_first = nullptr, *d_unit_count = nullptr, *d_unit_live = nullptr;
    uint32_t *d_adder_bitmap = nullptr, *d_class_bitmap = nullptr;
    unsigned long long *d_counters = nullptr, *d_sum_level = nullptr, *d_sum_power = nullptr, *d_win = nullptr;
    double *d_fsum_level = nullptr, *d_fsum_power = nullptr;
    uint32_t *d_bit_syndrome = nullptr;
    uint64_t *d_parity = nullptr, *d_tab_long = nullptr, *d_tab_short = nullptr;
    uint16_t *d_uc8_folded = nullptr;
    uint32_t *d_msg_pos = nullptr, *d_msg_limit = nullptr;
    uint16_t *d_msg_len = nullptr, *d_msg_skip = nullptr;
    unsigned long long *d_msg_sig = nullptr;
    int n_long = 0, n_short = 0;

    // pinned host
    PhaseRec *h_live = nullptr;
    unsigned long long *h_counters = nullptr, *h_sums = nullptr, *h_win = nullptr, *h_sig = nullptr;
    double *h_fsums = nullptr;
    uint32_t *h_total = nullptr;

    std::vector<SyndromeEntry> tab_long, tab_short;
    uint32_t valid_long = 0, valid_short = 0;

    Resolver resolver;
    std::vector<mgpu_msg> pending;
    mgpu_counters counters{};
    mgpu_timing timing{};
    uint64_t stream_pos = 0;   // samples consumed so far
    bool eof = false, have_tail = false;
};

#define HIPCHK(ctx, call)                                                                          \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                        \
            return e_ == hipErrorOutOfMemory ? MGPU_E_NOMEM : MGPU_E_HIP;                          \
        }                                                                                          \
    } while (0)

extern "C" {

void mgpu_config_defaults(struct mgpu_config *cfg) {
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
}

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

static int alloc_all(mgpu_ctx *c) {
    const mgpu_config &cfg = c->cfg;
    const uint64_t n = cfg.max_samples;
    c->cap_samples = n;
    c->cap_units = (n + kUnit - 1) / kUnit;
    c->cap_buffers = (n + cfg.buf_samples - 1) / cfg.buf_samples + 1;
    c->cap_pool = cfg.record_pool_records ? cfg.record_pool_records : n / 16 + 65536;
    if (c->cap_pool > 0xFFFFFFF0ull) c->cap_pool = 0xFFFFFFF0ull;
    c->cap_msgs = cfg.max_messages ? cfg.max_messages : n / 64 + 65536;
    const size_t bps = cfg.format == MGPU_FMT_UC8 ? 2 : 4;
    const uint64_t mag_len = (n + kTile - 1) / kTile * kTile + kTile + kHalo + 64;

    HIPCHK(c, hipMalloc(&c->d_iq, n * bps + 64));
    HIPCHK(c, hipMalloc(&c->d_mag, mag_len * sizeof(uint16_t)));
    HIPCHK(c, hipMemsetAsync(c->d_mag, 0, mag_len * sizeof(uint16_t), c->stream));
    HIPCHK(c, hipMalloc(&c->d_tail, kTrailing * sizeof(uint16_t)));
    HIPCHK(c, hipMalloc(&c->d_pool, c->cap_pool * sizeof(PhaseRec)));
    HIPCHK(c, hipMalloc(&c->d_live, c->cap_pool * sizeof(PhaseRec)));
    HIPCHK(c, hipMalloc(&c->d_pool_used, 64));
    HIPCHK(c, hipMalloc(&c->d_unit_first, (c->cap_units + 1) * sizeof(uint32_t)));
    HIPCHK(c, hipMalloc(&c->d_unit_count, (c->cap_units + 1) * sizeof(uint32_t)));
    HIPCHK(c, hipMalloc(&c->d_unit_live, (c->cap_units + 2) * sizeof(uint32_t)));
    HIPCHK(c, hipMalloc(&c->d_adder_bitmap, (1u << 24) / 8));
    HIPCHK(c, hipMemsetAsync(c->d_adder_bitmap, 0, (1u << 24) / 8, c->stream));
    HIPCHK(c, hipMalloc(&c->d_class_bitmap, (mag_len / 32 + 64) * sizeof(uint32_t)));
    HIPCHK(c, hipMalloc(&c->d_counters, CNT_NUM * sizeof(unsigned long long)));
    HIPCHK(c, hipMalloc(&c->d_sum_level, c->cap_buffers * sizeof(unsigned long long)));
    HIPCHK(c, hipMalloc(&c->d_sum_power, c->cap_buffers * sizeof(unsigned long long)));
    HIPCHK(c, hipMalloc(&c->d_fsum_level, c->cap_buffers * sizeof(double)));
    HIPCHK(c, hipMalloc(&c->d_fsum_power, c->cap_buffers * sizeof(double)));
    HIPCHK(c, hipMalloc(&c->d_win, 8 * sizeof(unsigned long long)));
    HIPCHK(c, hipMalloc(&c->d_msg_pos, c->cap_msgs * sizeof(uint32_t)));
    HIPCHK(c, hipMalloc(&c->d_msg_limit, c->cap_msgs * sizeof(uint32_t)));
    HIPCHK(c, hipMalloc(&c->d_msg_len, c->cap_msgs * sizeof(uint16_t)));
    HIPCHK(c, hipMalloc(&c->d_msg_skip, c->cap_msgs * sizeof(uint16_t)));
    HIPCHK(c, hipMalloc(&c->d_msg_sig, c->cap_msgs * sizeof(unsigned long long)));

    HIPCHK(c, hipHostMalloc(&c->h_live, c->cap_pool * sizeof(PhaseRec)));
    HIPCHK(c, hipHostMalloc(&c->h_counters, CNT_NUM * sizeof(unsigned long long)));
    HIPCHK(c, hipHostMalloc(&c->h_sums, 2 * c->cap_buffers * sizeof(unsigned long long)));
    HIPCHK(c, hipHostMalloc(&c->h_fsums, 2 * c->cap_buffers * sizeof(double)));
    HIPCHK(c, hipHostMalloc(&c->h_win, 8 * sizeof(unsigned long long)));
    HIPCHK(c, hipHostMalloc(&c->h_sig, c->cap_msgs * sizeof(unsigned long long)));
    HIPCHK(c, hipHostMalloc(&c->h_total, 64));

    // constant tables
    const CrcTables &crc = crc_tables();
    HIPCHK(c, hipMalloc(&c->d_bit_syndrome, 112 * sizeof(uint32_t)));
    HIPCHK(c, hipMemcpy(c->d_bit_syndrome, crc.bit_syndrome, 112 * sizeof(uint32_t), hipMemcpyHostToDevice));
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
        cfg->max_samples == 0 || cfg->max_samples > 0xF0000000ull || cfg->format < 0 || cfg->format > 2 ||
        cfg->nfix_crc < 0 || cfg->nfix_crc > 2)
        return MGPU_E_INVAL;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || cfg->device >= ndev) return MGPU_E_NODEVICE;
    mgpu_ctx *c = new (std::nothrow) mgpu_ctx();
    if (!c) return MGPU_E_NOMEM;
    c->cfg = *cfg;
    if (hipSetDevice(cfg->device) != hipSuccess) { delete c; return MGPU_E_NODEVICE; }
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return MGPU_E_HIP; }
    for (auto &e : c->ev)
        if (hipEventCreate(&e) != hipSuccess) { mgpu_destroy(c); return MGPU_E_HIP; }
    // valid_df_*_bitset, init_bitsets() demod_2400.c:112-128 (ENABLE_DF24 off, readsb.h:303)
    c->valid_short = (1u << 0) | (1u << 4) | (1u << 5) | (1u << 11);
    c->valid_long = (1u << 16) | (1u << 17) | (1u << 18) | (1u << 20) | (1u << 21);
    if (cfg->fixDF && cfg->nfix_crc)
        for (int b = 0; b < 5; ++b) c->valid_long |= 1u << (17 ^ (1 << b));
    int rc = alloc_all(c);
    if (rc != MGPU_OK) {
        std::fprintf(stderr, "mgpu_create: %s (%s)\n", mgpu_strerror(rc), c->err.c_str());
        mgpu_destroy(c);
        return rc;
    }
    c->resolver.reset(cfg->startup_time_ms);
    *out = c;
    return MGPU_OK;
}

void mgpu_destroy(mgpu_ctx *c) {
    if (!c) return;
    hipSetDevice(c->cfg.device);
    if (c->stream) hipStreamSynchronize(c->stream);
    void *dev[] = {c->d_iq, c->d_mag, c->d_tail, c->d_pool, c->d_live, c->d_pool_used, c->d_unit_first, c->d_unit_count,
                   c->d_unit_live, c->d_adder_bitmap, c->d_class_bitmap, c->d_counters, c->d_sum_level, c->d_sum_power,
                   c->d_fsum_level, c->d_fsum_power, c->d_win, c->d_msg_pos, c->d_msg_limit, c->d_msg_len, c->d_msg_skip,
                   c->d_msg_sig, c->d_bit_syndrome, c->d_parity, c->d_tab_long, c->d_tab_short, c->d_uc8_folded};
    for (void *p : dev)
        if (p) hipFree(p);
    void *host[] = {c->h_live, c->h_counters, c->h_sums, c->h_fsums, c->h_win, c->h_sig, c->h_total};
    for (void *p : host)
        if (p) hipHostFree(p);
    for (auto &e : c->ev)
        if (e) hipEventDestroy(e);
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
}

int mgpu_reset(mgpu_ctx *c) {
    if (!c) return MGPU_E_INVAL;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    c->resolver.reset(c->cfg.startup_time_ms);
    c->pending.clear();
    std::memset(&c->counters, 0, sizeof(c->counters));
    std::memset(&c->timing, 0, sizeof(c->timing));
    c->stream_pos = 0;
    c->eof = false;
    c->have_tail = false;
    HIPCHK(c, hipMemsetAsync(c->d_adder_bitmap, 0, (1u << 24) / 8, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return MGPU_OK;
}

// The pipeline behind every feed.  `have_mag`: d_mag[0 .. 326+n) already holds the buffer
// (mgpu_demod_mag_buf); otherwise d_iq holds n samples to convert.
static int run_feed(mgpu_ctx *c, uint64_t n, bool have_mag, const std::vector<BufferClock> &buffers,
                    const double *given_mean_power, float h2d_ms) {
    const mgpu_config &cfg = c->cfg;
    const double t_start = wall_ms();
    const uint32_t nbuf = (uint32_t) buffers.size();
    const uint32_t nunits = (uint32_t) ((n + kUnit - 1) / kUnit);
    hipStream_t s = c->stream;
    mgpu_timing tm{};
    tm.h2d_ms = h2d_ms;

    HIPCHK(c, hipMemsetAsync(c->d_counters, 0, CNT_NUM * sizeof(unsigned long long), s));
    HIPCHK(c, hipMemsetAsync(c->d_pool_used, 0, sizeof(uint32_t), s));
    HIPCHK(c, hipMemsetAsync(c->d_win, 0, 8 * sizeof(unsigned long long), s));
    HIPCHK(c, hipMemsetAsync(c->d_sum_level, 0, nbuf * sizeof(unsigned long long), s));
    HIPCHK(c, hipMemsetAsync(c->d_sum_power, 0, nbuf * sizeof(unsigned long long), s));
    HIPCHK(c, hipMemsetAsync(c->d_fsum_level, 0, nbuf * sizeof(double), s));
    HIPCHK(c, hipMemsetAsync(c->d_fsum_power, 0, nbuf * sizeof(double), s));

    // ---- convert ----
    HIPCHK(c, hipEventRecord(c->ev[0], s));
    if (!have_mag) {
        if (c->have_tail) HIPCHK(c, hipMemcpyAsync(c->d_mag, c->d_tail, kTrailing * sizeof(uint16_t), hipMemcpyDeviceToDevice, s));
        else HIPCHK(c, hipMemsetAsync(c->d_mag, 0, kTrailing * sizeof(uint16_t), s));
        ConvertParams cp{};
        cp.iq = c->d_iq; cp.mag = c->d_mag; cp.n = n; cp.buf_samples = cfg.buf_samples;
        cp.uc8_folded = c->d_uc8_folded;
        cp.sum_level = c->d_sum_level; cp.sum_power = c->d_sum_power;
        cp.fsum_level = c->d_fsum_level; cp.fsum_power = c->d_fsum_power;
        launch_convert(cfg.format, cp, s);
    }
    HIPCHK(c, hipEventRecord(c->ev[1], s));

    // ---- sweep + slice ----
    SweepParams sp{};
    sp.mag = c->d_mag; sp.n = n; sp.thr = cfg.preamble_threshold;
    sp.valid_long = c->valid_long; sp.valid_short = c->valid_short;
    sp.fix_df = (cfg.fixDF && cfg.nfix_crc) ? 1 : 0;
    sp.bit_syndrome = c->d_bit_syndrome; sp.parity = c->d_parity;
    sp.tab_long = c->d_tab_long; sp.tab_short = c->d_tab_short; sp.n_long = c->n_long; sp.n_short = c->n_short;
    sp.pool = c->d_pool; sp.pool_cap = (uint32_t) c->cap_pool; sp.pool_used = c->d_pool_used;
    sp.unit_first = c->d_unit_first; sp.unit_count = c->d_unit_count; sp.nunits = nunits;
    sp.adder_bitmap = c->d_adder_bitmap; sp.class_bitmap = c->d_class_bitmap; sp.counters = c->d_counters;
    launch_sweep_slice(sp, s);
    HIPCHK(c, hipEventRecord(c->ev[2], s));

    // ---- pre-screen ----
    launch_prescreen(c->d_pool, c->d_unit_first, nunits, c->d_adder_bitmap, c->d_unit_live, c->d_live, c->d_counters, s);
    HIPCHK(c, hipEventRecord(c->ev[3], s));
    HIPCHK(c, hipMemcpyAsync(c->h_counters, c->d_counters, CNT_NUM * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    if (nunits) HIPCHK(c, hipMemcpyAsync(c->h_total, c->d_unit_live + nunits, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    else c->h_total[0] = 0;
    if (!have_mag) {
        HIPCHK(c, hipMemcpyAsync(c->h_sums, c->d_sum_level, nbuf * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipMemcpyAsync(c->h_sums + nbuf, c->d_sum_power, nbuf * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipMemcpyAsync(c->h_fsums, c->d_fsum_level, nbuf * sizeof(double), hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipMemcpyAsync(c->h_fsums + nbuf, c->d_fsum_power, nbuf * sizeof(double), hipMemcpyDeviceToHost, s));
    }
    HIPCHK(c, hipStreamSynchronize(s));
    if (c->h_counters[CNT_POOL_OVERFLOW]) {
        c->err = "record pool overflow: recreate the context with a larger record_pool_records";
        return MGPU_E_OVERFLOW;
    }
    const uint64_t nlive = c->h_total[0];
    HIPCHK(c, hipEventRecord(c->ev[4], s));
    if (nlive) HIPCHK(c, hipMemcpyAsync(c->h_live, c->d_live, nlive * sizeof(PhaseRec), hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipEventRecord(c->ev[5], s));
    HIPCHK(c, hipStreamSynchronize(s));

    // ---- ordered walk (host) ----
    const double t_res0 = wall_ms();
    std::vector<mgpu_msg> msgs;
    std::vector<uint32_t> mpos, mlimit;
    std::vector<uint16_t> mskip;
    ResolveCounts rc;
    c->resolver.walk(c->h_live, nlive, buffers, msgs, mpos, mskip, mlimit, rc);
    tm.resolve_ms = (float) (wall_ms() - t_res0);
    const uint32_t nmsg = (uint32_t) msgs.size();
    if (nmsg > c->cap_msgs) { c->err = "max_messages exceeded"; return MGPU_E_OVERFLOW; }

    // ---- signal power + window statistics ----
    HIPCHK(c, hipEventRecord(c->ev[6], s));
    if (nmsg) {
        std::vector<uint16_t> mlen(nmsg);
        for (uint32_t i = 0; i < nmsg; ++i) mlen[i] = msgs[i].sig_len;
        HIPCHK(c, hipMemcpyAsync(c->d_msg_pos, mpos.data(), nmsg * sizeof(uint32_t), hipMemcpyHostToDevice, s));
        HIPCHK(c, hipMemcpyAsync(c->d_msg_limit, mlimit.data(), nmsg * sizeof(uint32_t), hipMemcpyHostToDevice, s));
        HIPCHK(c, hipMemcpyAsync(c->d_msg_len, mlen.data(), nmsg * sizeof(uint16_t), hipMemcpyHostToDevice, s));
        HIPCHK(c, hipMemcpyAsync(c->d_msg_skip, mskip.data(), nmsg * sizeof(uint16_t), hipMemcpyHostToDevice, s));
        launch_signal_power(c->d_mag, c->d_msg_pos, c->d_msg_len, nmsg, c->d_msg_sig, s);
        launch_window_stats(c->d_mag, n, cfg.preamble_threshold, c->d_class_bitmap, c->d_msg_pos, c->d_msg_skip,
                            c->d_msg_limit, nmsg, c->d_win, s);
        HIPCHK(c, hipMemcpyAsync(c->h_sig, c->d_msg_sig, nmsg * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipStreamSynchronize(s));   // mlen & co. are pageable: keep them alive until copied
    }
    HIPCHK(c, hipMemcpyAsync(c->h_win, c->d_win, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    // carry the last 326 magnitudes into the next feed (sdr_ifile.c:209-213)
    if (n >= (uint64_t) kTrailing || have_mag) {
        HIPCHK(c, hipMemcpyAsync(c->d_tail, c->d_mag + n, kTrailing * sizeof(uint16_t), hipMemcpyDeviceToDevice, s));
        c->have_tail = true;
    } else {
        c->have_tail = false;   // lastbuf->length < trailing_samples -> zeros
    }
    HIPCHK(c, hipEventRecord(c->ev[7], s));
    HIPCHK(c, hipStreamSynchronize(s));

    // ---- counters (Modes.stats_current) ----
    mgpu_counters &k = c->counters;
    const unsigned long long *hc = c->h_counters, *hw = c->h_win;
    const uint64_t C = hc[CNT_CANDIDATES], U = hc[CNT_CLASS_COND], R = hc[CNT_CLASS_UNCOND];
    const uint64_t cW = hw[0], uW = hw[4];
    k.demod_preambles += C - cW;
    k.demod_preamblePhase[0] += hc[CNT_PHASE0 + 0] - hw[1];
    k.demod_preamblePhase[1] += hc[CNT_PHASE0 + 1] - hw[1];
    k.demod_preamblePhase[2] += hc[CNT_PHASE0 + 2] - hw[2];
    k.demod_preamblePhase[3] += hc[CNT_PHASE0 + 3] - hw[2];
    k.demod_preamblePhase[4] += hc[CNT_PHASE0 + 4] - hw[3];
    // candidates without any record score -2 for sure; those hidden inside skip windows are not counted
    k.demod_rejected_bad += (C - U - R) - (cW - uW - rc.skipped_uncond_groups) + rc.rejected_bad;
    // conditional-only candidates: dead ones (address can never be known) + the visited live ones the walk rejected
    k.demod_rejected_unknown_icao += rc.rejected_unknown + (U - rc.visited_cond_groups - uW);
    for (int i = 0; i < 3; ++i) k.demod_accepted[i] += rc.accepted[i];
    for (int i = 0; i < 5; ++i) k.demod_bestPhase[i] += rc.best_phase[i];

    // per-message signal level, per-buffer noise power (demod_2400.c:436-457, 474-479)
    uint32_t mi = 0;
    for (uint32_t b = 0; b < nbuf; ++b) {
        const BufferClock &bc = buffers[b];
        uint64_t sum_scaled = 0;
        while (mi < nmsg && mpos[mi] < (uint64_t) bc.first + bc.length) {
            mgpu_msg &m = msgs[mi];
            m.sig_sumsq = c->h_sig[mi];
            const double signal_power = (double) m.sig_sumsq / 65535.0 / 65535.0;
            const double level = signal_power / m.sig_len;
            k.signal_power_sum += signal_power;
            k.signal_power_count += m.sig_len;
            sum_scaled += m.sig_sumsq;
            if (level > k.peak_signal_power) k.peak_signal_power = level;
            if (level > 0.50119) k.strong_signal_count++;
            ++mi;
        }
        double mean_power;
        if (given_mean_power) mean_power = given_mean_power[b];
        else if (cfg.format == MGPU_FMT_UC8) mean_power = (double) c->h_sums[nbuf + b] / 65535.0 / 65535.0 / bc.length;   // convert.c:105-107
        else mean_power = c->h_fsums[nbuf + b] / bc.length;
        const double sum_signal_power = (double) sum_scaled / 65535.0 / 65535.0;
        k.noise_power_sum += (mean_power * bc.length - sum_signal_power);
        k.noise_power_count += bc.length;
        k.samples_processed += bc.length;
        k.samples_lost += cfg.buf_samples - bc.length;        // readsb.c:886
        k.nbuffers++;
    }
    k.nflips = c->resolver.nflips();
    c->pending.insert(c->pending.end(), msgs.begin(), msgs.end());

    hipEventElapsedTime(&tm.convert_ms, c->ev[0], c->ev[1]);
    hipEventElapsedTime(&tm.sweep_ms, c->ev[1], c->ev[2]);
    hipEventElapsedTime(&tm.prescreen_ms, c->ev[2], c->ev[3]);
    hipEventElapsedTime(&tm.d2h_ms, c->ev[4], c->ev[5]);
    hipEventElapsedTime(&tm.sigpower_ms, c->ev[6], c->ev[7]);
    tm.total_ms = (float) (wall_ms() - t_start) + h2d_ms;
    tm.n_candidates = C;
    tm.n_records = hc[CNT_RECORDS];
    tm.n_live_records = nlive;
    tm.n_messages = nmsg;
    c->timing = tm;
    return MGPU_OK;
}

// buffer grid of ifileRun for `n` samples continuing at stream position `pos0` (sdr_ifile.c:194-241)
static std::vector<BufferClock> ifile_grid(const mgpu_ctx *c, uint64_t pos0, uint64_t n) {
    std::vector<BufferClock> v;
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
    return v;
}

static int feed_common(mgpu_ctx *c, const void *src, bool src_is_device, uint64_t n) {
    if (!c || (!src && n)) return MGPU_E_INVAL;
    if (n == 0) return MGPU_OK;
    if (c->eof) return MGPU_E_EOF;
    if (n > c->cap_samples) return MGPU_E_CAPACITY;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    const size_t bps = c->cfg.format == MGPU_FMT_UC8 ? 2 : 4;
    float h2d = 0.f;
    if (src_is_device) {
        if (src != c->d_iq) HIPCHK(c, hipMemcpyAsync(c->d_iq, src, n * bps, hipMemcpyDeviceToDevice, c->stream));
    } else {
        const double t0 = wall_ms();
        HIPCHK(c, hipMemcpyAsync(c->d_iq, src, n * bps, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        h2d = (float) (wall_ms() - t0);
    }
    const std::vector<BufferClock> grid = ifile_grid(c, c->stream_pos, n);
    int rc = run_feed(c, n, false, grid, nullptr, h2d);
    if (rc != MGPU_OK) return rc;
    c->stream_pos += n;
    if (n % c->cfg.buf_samples) c->eof = true;   // short read = end of file (sdr_ifile.c:223-237)
    return MGPU_OK;
}

int mgpu_feed_iq(mgpu_ctx *c, const void *iq_host, uint64_t nsamples) { return feed_common(c, iq_host, false, nsamples); }

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
    if (c->eof) return MGPU_OK;
    // file length an exact multiple of the buffer size: one more zero-length buffer, whose
    // converter call divides 0 by 0 (convert.c:101-107) -> noise_power_sum becomes NaN
    const int64_t st = (int64_t) c->stream_pos * 5;
    c->resolver.tick_empty(st / 12000 + c->cfg.startup_time_ms);
    c->counters.noise_power_sum += std::numeric_limits<double>::quiet_NaN();
    c->counters.samples_lost += c->cfg.buf_samples;
    c->counters.nbuffers++;
    c->counters.nflips = c->resolver.nflips();
    c->eof = true;
    return MGPU_OK;
}

int mgpu_collect(mgpu_ctx *c, struct mgpu_msg *out, uint64_t cap, uint64_t *n, struct mgpu_counters *counters) {
    if (!c || (!out && cap)) return MGPU_E_INVAL;
    uint64_t k = c->pending.size() < cap ? c->pending.size() : cap;
    if (k) std::memcpy(out, c->pending.data(), k * sizeof(mgpu_msg));
    c->pending.erase(c->pending.begin(), c->pending.begin() + (long) k);
    if (n) *n = k;
    if (counters) *counters = c->counters;
    return MGPU_OK;
}

uint64_t mgpu_pending_messages(mgpu_ctx *c) { return c ? c->pending.size() : 0; }

int mgpu_last_timing(mgpu_ctx *c, struct mgpu_timing *t) {
    if (!c || !t) return MGPU_E_INVAL;
    *t = c->timing;
    return MGPU_OK;
}

int mgpu_convert(mgpu_ctx *c, const void *iq_host, uint16_t *mag_host, uint32_t n, double *out_mean_level, double *out_mean_power) {
    if (!c || !iq_host || !mag_host) return MGPU_E_INVAL;
    if (n > c->cap_samples) return MGPU_E_CAPACITY;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    const size_t bps = c->cfg.format == MGPU_FMT_UC8 ? 2 : 4;
    hipStream_t s = c->stream;
    double ml = std::numeric_limits<double>::quiet_NaN(), mp = ml;   // 0/0 for n == 0, as the reference
    if (n) {
        HIPCHK(c, hipMemcpyAsync(c->d_iq, iq_host, (size_t) n * bps, hipMemcpyHostToDevice, s));
        HIPCHK(c, hipMemsetAsync(c->d_sum_level, 0, c->cap_buffers * sizeof(unsigned long long), s));
        HIPCHK(c, hipMemsetAsync(c->d_sum_power, 0, c->cap_buffers * sizeof(unsigned long long), s));
        HIPCHK(c, hipMemsetAsync(c->d_fsum_level, 0, c->cap_buffers * sizeof(double), s));
        HIPCHK(c, hipMemsetAsync(c->d_fsum_power, 0, c->cap_buffers * sizeof(double), s));
        ConvertParams cp{};
        cp.iq = c->d_iq; cp.mag = c->d_mag; cp.n = n;
        cp.buf_samples = 0x80000000u;   // one accumulation bucket for the whole call
        cp.uc8_folded = c->d_uc8_folded;
        cp.sum_level = c->d_sum_level; cp.sum_power = c->d_sum_power;
        cp.fsum_level = c->d_fsum_level; cp.fsum_power = c->d_fsum_power;
        launch_convert(c->cfg.format, cp, s);
        HIPCHK(c, hipMemcpyAsync(mag_host, c->d_mag + kTrailing, (size_t) n * sizeof(uint16_t), hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipMemcpyAsync(c->h_sums, c->d_sum_level, sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipMemcpyAsync(c->h_sums + 1, c->d_sum_power, sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipMemcpyAsync(c->h_fsums, c->d_fsum_level, sizeof(double), hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipMemcpyAsync(c->h_fsums + 1, c->d_fsum_power, sizeof(double), hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipStreamSynchronize(s));
        if (c->cfg.format == MGPU_FMT_UC8) {
            ml = (double) c->h_sums[0] / 65536.0 / n;              // convert.c:101-103 (sic, 65536)
            mp = (double) c->h_sums[1] / 65535.0 / 65535.0 / n;    // convert.c:105-107
        } else {
            ml = c->h_fsums[0] / n;
            mp = c->h_fsums[1] / n;
        }
    }
    if (out_mean_level) *out_mean_level = ml;
    if (out_mean_power) *out_mean_power = mp;
    return MGPU_OK;
}

int mgpu_demod_mag_buf(mgpu_ctx *c, const uint16_t *data, uint32_t length, int64_t sampleTimestamp, int64_t sysTimestamp,
                       double mean_power, uint32_t dropped) {
    if (!c || !data) return MGPU_E_INVAL;
    (void) dropped;   // raising the threshold after drops (demod_2400.c:335-338) is the caller's cfg.preamble_threshold
    if (length > c->cap_samples) return MGPU_E_CAPACITY;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    std::vector<BufferClock> grid(1);
    grid[0].first = 0; grid[0].length = length;
    grid[0].sampleTimestamp = sampleTimestamp; grid[0].sysTimestamp = sysTimestamp;
    if (length == 0) {
        c->resolver.tick_empty(sysTimestamp);
        c->counters.noise_power_sum += mean_power * 0.0;
        c->counters.samples_lost += c->cfg.buf_samples;
        c->counters.nbuffers++;
        c->counters.nflips = c->resolver.nflips();
        return MGPU_OK;
    }
    const double t0 = wall_ms();
    HIPCHK(c, hipMemcpyAsync(c->d_mag, data, ((size_t) length + kTrailing) * sizeof(uint16_t), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const float h2d = (float) (wall_ms() - t0);
    int rc = run_feed(c, length, true, grid, &mean_power, h2d);
    if (rc == MGPU_OK) c->stream_pos += length;
    return rc;
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
