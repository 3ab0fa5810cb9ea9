/* demod_gpu_wrap.c — the adapter of INTEGRATION.md §2 as a real file that compiles against the reference's own headers,
 * in the form that needs NO change to the reference's sources: linked with
 *     -Wl,--wrap=demodulate2400 -Wl,--wrap=demodulate2400AC -Wl,--wrap=icaoFilterExpire -Wl,--wrap=icaoFilterAdd
 * every call the decode thread makes (readsb.c:871-874) lands here instead of in demod_2400.c, and the GPU library does the
 * work.  (Inside a readsb tree one would rename the two functions and switch on a --gpu option instead; the body is the same.)
 *
 * Per buffer: mgpu_demod_mag_buf[_ac]() on the struct mag_buf the SDR thread filled, then for every returned message exactly
 * what demodulate2400 / demodulate2400AC do after they have settled on a frame (demod_2400.c:401-471, 742-758):
 * netGetMM, timestamps, score, decodeModesMessage / decodeModeAMessage on the frame as sliced, signalLevel, netUseMessage —
 * so decodeModesMessage keeps readsb's own ICAO filter in step (same adds, same order) and everything behind netUseMessage
 * (tracking, beast / raw / SBS output, statistics) sees what it always saw.  The demodulator counters of struct stats are
 * advanced by the library's counter deltas.
 *
 * The ICAO filter clock stays the host's: the library runs with MGPU_FILTER_CLOCK_EXTERNAL and every icaoFilterExpire() /
 * icaoFilterAdd() the program performs (backgroundTasks, readsb.c:1227-1231; decodeModesMessage on any input,
 * mode_s.c:766-779) is forwarded, so the two filters flip between the same two buffers whichever of the program's two
 * start-up orders occurs (first flip before or after buffer 0, readsb.c:857-902) and whatever clock drives it (synthetic
 * for ifile, wall clock for a live SDR).
 *
 * Built by `make -C oracle full_gpu` against /root/reference's headers and objects (outputs under oracle/_ref/full/).
 */
#include "readsb.h"
#include "modes_gpu.h"

static mgpu_ctx *gpu;
static struct mgpu_counters seen;          /* the library's counters are cumulative: remember what was already added */
static int failed;

/* filter operations the program performed before the library context existed (modesInit's icaoFilterAdd(show_only), a
 * first backgroundTasks pass that found no buffer): replayed in order when the context is created */
static struct early_op { uint32_t addr; int expire; } *early;
static size_t n_early, cap_early;

static void early_push(int expire, uint32_t addr) {
    if (n_early == cap_early) {
        cap_early = cap_early ? 2 * cap_early : 64;
        early = realloc(early, cap_early * sizeof(*early));
        if (!early) { fprintf(stderr, "<3>GPU demodulator: out of memory\n"); exit(1); }
    }
    early[n_early].expire = expire;
    early[n_early++].addr = addr;
}

void __real_icaoFilterExpire(void);
void __real_icaoFilterAdd(uint32_t addr);

void __wrap_icaoFilterExpire(void) {
    __real_icaoFilterExpire();
    if (gpu) (void) mgpu_filter_expire(gpu);
    else if (!failed) early_push(1, 0);
}

void __wrap_icaoFilterAdd(uint32_t addr) {
    __real_icaoFilterAdd(addr);
    if (gpu) (void) mgpu_filter_add(gpu, addr);
    else if (!failed) early_push(0, addr);
}

static void gpu_close(void) {                               /* at exit: stop the library's pipeline threads before the runtime unloads */
    if (gpu) mgpu_destroy(gpu);
    gpu = NULL;
}

static void gpu_open(void) {
    struct mgpu_config cfg;
    mgpu_config_defaults(&cfg);
    const char *dev = getenv("READSB_GPU_DEVICE");          /* a --gpu-device option in a real tree */
    cfg.device = dev ? atoi(dev) : 0;
    cfg.nfix_crc = Modes.nfix_crc;
    cfg.fixDF = Modes.fixDF;
    cfg.preamble_threshold = Modes.preambleThreshold;
    cfg.mode_ac = Modes.mode_ac ? 1 : 0;
    cfg.buf_samples = Modes.sdr_buf_samples;
    cfg.trailing_samples = Modes.trailing_samples;
    cfg.max_samples = Modes.sdr_buf_samples;                /* one struct mag_buf per call */
    cfg.startup_time_ms = Modes.startup_time;
    cfg.filter_clock = MGPU_FILTER_CLOCK_EXTERNAL;          /* the program's own backgroundTasks decides, see above */
    const int rc = mgpu_create(&cfg, &gpu);
    if (rc != MGPU_OK) {                                    /* loud, like a failing sdrOpen() (readsb.c:501-506): no CPU fallback */
        fprintf(stderr, "<3>GPU demodulator: %s\n", mgpu_strerror(rc));
        gpu = NULL;
        failed = 1;
        setExit(2);
        return;
    }
    atexit(gpu_close);
    int flips = 0;
    for (size_t i = 0; i < n_early; i++) {
        if (early[i].expire) { (void) mgpu_filter_expire(gpu); flips++; }
        else (void) mgpu_filter_add(gpu, early[i].addr);
    }
    fprintf(stderr, "GPU demodulator: following the program's ICAO filter clock (first flip %s buffer 0)\n", flips ? "before" : "after");
    free(early);
    early = NULL;
    n_early = cap_early = 0;
}

static void add_counter_deltas(const struct mgpu_counters *c) {
    struct stats *s = &Modes.stats_current;                 /* the fields demodulate2400 / demodulate2400AC advance themselves */
    s->demod_preambles += c->demod_preambles - seen.demod_preambles;
    s->demod_rejected_bad += c->demod_rejected_bad - seen.demod_rejected_bad;
    s->demod_rejected_unknown_icao += c->demod_rejected_unknown_icao - seen.demod_rejected_unknown_icao;
    for (int i = 0; i < 3; ++i) s->demod_accepted[i] += c->demod_accepted[i] - seen.demod_accepted[i];
    for (int i = 0; i < 5; ++i) {
        s->demod_preamblePhase[i] += c->demod_preamblePhase[i] - seen.demod_preamblePhase[i];
        s->demod_bestPhase[i] += c->demod_bestPhase[i] - seen.demod_bestPhase[i];
    }
    s->demod_modeac += c->demod_modeac - seen.demod_modeac;
    s->strong_signal_count += c->strong_signal_count - seen.strong_signal_count;
    s->signal_power_sum += c->signal_power_sum - seen.signal_power_sum;
    s->signal_power_count += c->signal_power_count - seen.signal_power_count;
    s->noise_power_sum += c->noise_power_sum - seen.noise_power_sum;
    s->noise_power_count += c->noise_power_count - seen.noise_power_count;
    if (c->peak_signal_power > s->peak_signal_power) s->peak_signal_power = c->peak_signal_power;
    seen = *c;
}

void __wrap_demodulate2400(struct mag_buf *mag) {
    if (!gpu && !failed) gpu_open();
    if (!gpu) return;
    if (Modes.sdr_type == SDR_IFILE && Modes.synthetic_now)
        Modes.synthetic_now = mag->sysTimestamp;                                   /* demod_2400.c:283-285 */
    const uint32_t dropped = Modes.stats_15min.samples_dropped ? 1 : 0;            /* what demod_2400.c:335-338 looks at */
    const int rc = Modes.mode_ac
        ? mgpu_demod_mag_buf_ac(gpu, mag->data, mag->length, mag->sampleTimestamp, mag->sysTimestamp, mag->mean_level, mag->mean_power, dropped)
        : mgpu_demod_mag_buf(gpu, mag->data, mag->length, mag->sampleTimestamp, mag->sysTimestamp, mag->mean_power, dropped);
    if (rc != MGPU_OK) {
        fprintf(stderr, "<3>GPU demodulator: %s (%s)\n", mgpu_strerror(rc), mgpu_last_error(gpu));
        failed = 1;
        setExit(2);
        return;
    }
    static struct mgpu_msg batch[4096];
    struct mgpu_counters c = seen;                                                 /* a failing first collect adds nothing */
    uint64_t n = 0;
    do {
        if (mgpu_collect(gpu, batch, 4096, &n, &c) != MGPU_OK) { c = seen; break; }
        for (uint64_t i = 0; i < n; i++) {
            const struct mgpu_msg *m = &batch[i];
            struct modesMessage *mm = netGetMM(&Modes.netMessageBuffer[0]);
            mm->timestamp = m->timestamp;
            mm->sysTimestamp = m->sysTimestamp;
            if (m->msgbits == 16) {                                                /* a Mode A/C reply: demod_2400.c:742-754 */
                decodeModeAMessage(mm, (m->msg[0] << 8) | m->msg[1]);
                netUseMessage(mm);
                continue;
            }
            if (Modes.sdr_type == SDR_IFILE && Modes.synthetic_now)
                Modes.synthetic_now = mm->sysTimestamp;                            /* demod_2400.c:412-414 */
            mm->score = m->score;
            memcpy(mm->msg, m->raw, MODES_LONG_MSG_BYTES);                         /* the frame as sliced, :420 */
            if (decodeModesMessage(mm) < 0) {                                      /* repeats the repair; adds to readsb's own filter */
                fprintf(stderr, "<3>GPU demodulator: readsb's decodeModesMessage rejected an accepted frame (filters out of step)\n");
                continue;
            }
            mm->signalLevel = mgpu_msg_signal_level(m);                            /* :447-448 */
            netUseMessage(mm);
        }
    } while (n == 4096);
    add_counter_deltas(&c);
    netDrainMessageBuffers();                                                      /* :481 */
}

/* demodulate2400AC(buf) follows demodulate2400(buf) when Modes.mode_ac is set (readsb.c:872-874): the call above already
 * delivered the buffer's Mode A/C replies behind its Mode S messages, in the reference's order */
void __wrap_demodulate2400AC(struct mag_buf *mag) { (void) mag; }

/* ---- the converter side of the boundary: iq_convert_fn (convert.h:34-39) ------------------------------------------------
 * With READSB_GPU_CONVERT=1 in the environment and `-Wl,--wrap=init_converter`, the SDR reader thread's converter
 * (sdr_ifile.c:156,238; sdr_rtlsdr.c:271) is mgpu_convert() on its own context: IQ block up, magnitudes and the two means
 * back.  A round trip over PCIe per buffer — there for the completeness of the drop-in (the bulk path hands the library IQ
 * and never brings magnitudes back), off by default. */
iq_convert_fn __real_init_converter(input_format_t format, double sample_rate, int filter_dc, struct converter_state **out_state);

static mgpu_ctx *conv_gpu;

static void gpu_convert_close(void) {
    if (conv_gpu) mgpu_destroy(conv_gpu);
    conv_gpu = NULL;
}

static void gpu_convert(void *iq_data, uint16_t *mag_data, unsigned nsamples, struct converter_state *state,
                        double *out_mean_level, double *out_mean_power) {
    (void) state;
    const int rc = mgpu_convert(conv_gpu, iq_data, mag_data, nsamples, out_mean_level, out_mean_power);
    if (rc != MGPU_OK) {
        fprintf(stderr, "<3>GPU converter: %s (%s)\n", mgpu_strerror(rc), mgpu_last_error(conv_gpu));
        setExit(2);
    }
}

iq_convert_fn __wrap_init_converter(input_format_t format, double sample_rate, int filter_dc, struct converter_state **out_state) {
    const char *on = getenv("READSB_GPU_CONVERT");
    if (!on || !atoi(on) || filter_dc)                       /* the --dcfilter converters are a serial IIR: not offered */
        return __real_init_converter(format, sample_rate, filter_dc, out_state);
    struct mgpu_config cfg;
    mgpu_config_defaults(&cfg);
    const char *dev = getenv("READSB_GPU_DEVICE");
    cfg.device = dev ? atoi(dev) : 0;
    cfg.format = format == INPUT_UC8 ? MGPU_FMT_UC8 : format == INPUT_SC16 ? MGPU_FMT_SC16 : MGPU_FMT_SC16Q11;
    cfg.max_samples = Modes.sdr_buf_samples ? Modes.sdr_buf_samples : 131072;   /* one buffer per call (sdr_ifile.c:238) */
    const int rc = mgpu_create(&cfg, &conv_gpu);
    if (rc != MGPU_OK) {
        fprintf(stderr, "<3>GPU converter: %s\n", mgpu_strerror(rc));
        conv_gpu = NULL;
        return NULL;                                        /* init_converter's own failure value: the plugin's open() fails */
    }
    atexit(gpu_convert_close);
    *out_state = NULL;
    fprintf(stderr, "init_converter: using the GPU library\n");
    return gpu_convert;
}

