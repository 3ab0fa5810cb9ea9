/* ref_harness.c — drives the REFERENCE's own hot-path objects (convert.c, crc.c,
 * demod_2400.c, mode_s.c, icao_filter.c, comm_b.c, mode_ac.c, ais_charset.c,
 * compiled from /root/reference where they lie; see oracle/Makefile) through the
 * same buffer grid sdr_ifile.c produces, and records every message that reaches
 * netUseMessage().  TEST INFRASTRUCTURE ONLY (see oracle_io.h).
 *
 * What it reproduces (reference file:line):
 *   - ifileRun's buffer model: 131072-sample blocks, 326-sample overlap copied from
 *     the previous block, sampleTimestamp = sampleCounter*5, sysTimestamp =
 *     sampleTimestamp/12000 + startup_time, short final block, and one extra
 *     zero-length block when the file is an exact multiple (sdr_ifile.c:194-241)
 *   - decodeEntryPoint's per-buffer sequence: demodulate2400(buf) then
 *     backgroundTasks(now = mstime()) which flips the ICAO filter every 60 s of
 *     synthetic time, starting with next_flip = 0 (readsb.c:869-902, 1227-1231)
 *   - modesInit's order: modesChecksumInit(nfix), icaoFilterInit(),
 *     icaoFilterAdd(show_only = BADDR) (readsb.c:306-310)
 *
 * It supplies the eight symbols those objects leave undefined: Modes, setExit,
 * netGetMM, netUseMessage, netDrainMessageBuffers, receiveclock_ms_elapsed
 * (util.c:149), printACASInfoShort, sprint_uuid1.
 *
 * The reference keeps per-process static state (valid_df bitsets are initialised
 * once, demod_2400.c:276), so ONE configuration per process: use the CLI
 * (`ref_demod`), one process per run.
 */
#include "readsb.h"
#include "oracle_io.h"

struct _Modes Modes;

static struct modesMessage g_slot;
static struct messageBuffer g_mb;

static struct oracle_msg *g_out;
static uint64_t g_nout, g_capout;

void setExit(int arg) { Modes.exit = arg; }

int64_t receiveclock_ms_elapsed(int64_t t1, int64_t t2) { return (t2 - t1) / 12000U; }

void printACASInfoShort(uint32_t addr, unsigned char *MV, struct aircraft *a, struct modesMessage *mm, int64_t now) {
    (void) addr; (void) MV; (void) a; (void) mm; (void) now;
}

char *sprint_uuid1(uint64_t id1, char *p) { (void) id1; *p = 0; return p; }

struct modesMessage *netGetMM(struct messageBuffer *buf) {
    memset(&g_slot, 0, sizeof(g_slot));
    g_slot.messageBuffer = buf;
    return &g_slot;
}

void netUseMessage(struct modesMessage *mm) {
    if (g_nout == g_capout) {
        g_capout = g_capout ? g_capout * 2 : 65536;
        g_out = realloc(g_out, g_capout * sizeof(*g_out));
        if (!g_out) { fprintf(stderr, "ref_harness: out of memory\n"); exit(1); }
    }
    struct oracle_msg *o = &g_out[g_nout++];
    memset(o, 0, sizeof(*o));
    o->timestamp = mm->timestamp;
    o->sys_rel_ms = mm->sysTimestamp - Modes.startup_time;
    o->score = mm->score;
    o->correctedbits = mm->correctedbits;
    o->msgbits = mm->msgbits;
    o->msgtype = mm->msgtype;
    /* low 24 bits: the CRC/address stage's value; the later ES field decode may OR in
     * MODES_NON_ICAO_ADDRESS for some DF18 formats, which is outside the hot path */
    o->addr = mm->addr & 0xffffff;
    /* demodulate2400 memcpy's all 14 bytes of its scratch buffer (demod_2400.c:420); for a
     * 56-bit frame bytes 7..13 are leftovers of earlier slices, so only msgbits/8 bytes are kept */
    memcpy(o->msg, mm->msg, mm->msgbits / 8);
    memcpy(o->raw, mm->verbatim, mm->msgbits / 8);
    o->signalLevel = mm->signalLevel;
}

void netDrainMessageBuffers(void) { }

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + ts.tv_nsec * 1e-9;
}

static int g_mode_ac;   /* --modeac: demodulate2400AC after demodulate2400 on every buffer (readsb.c:871-874) */
void ref_set_mode_ac(int on) { g_mode_ac = on; }
static int g_flip_before;   /* the reference program's other start-up order: first icaoFilterExpire() before buffer 0 */
void ref_set_flip_before(int on) { g_flip_before = on; }

static struct converter_state *g_cstate;
static iq_convert_fn g_conv;
static int g_configured;

static int ref_configure(int format, int nfix, int fixdf, int thr) {
    if (g_configured) {
        fprintf(stderr, "ref_harness: one configuration per process\n");
        return -1;
    }
    g_configured = 1;
    memset(&Modes, 0, sizeof(Modes));
    Modes.nfix_crc = nfix;
    Modes.fixDF = fixdf;
    Modes.preambleThreshold = thr;
    Modes.sample_rate = 2400000.0;
    Modes.sdr_buf_size = 256 * 1024;
    Modes.sdr_buf_samples = Modes.sdr_buf_size / 2;
    Modes.trailing_samples = (unsigned) ((MODES_PREAMBLE_US + MODES_LONG_MSG_BITS + 16) * 1e-6 * Modes.sample_rate);
    Modes.sdr_type = SDR_IFILE;
    Modes.decodeThreads = 1;
    Modes.decode_all = 0;
    Modes.net_verbatim = 1;  /* keep the uncorrected bytes in mm->verbatim (mode_s.c:444) */
    Modes.show_only = BADDR;
    Modes.startup_time = ORACLE_STARTUP_MS;
    Modes.synthetic_now = Modes.startup_time;
    g_mb.msg = &g_slot; g_mb.len = 0; g_mb.alloc = 1 << 30;
    Modes.netMessageBuffer = &g_mb;

    modesChecksumInit(Modes.nfix_crc);
    icaoFilterInit();
    icaoFilterAdd(Modes.show_only);

    g_conv = init_converter((input_format_t) format, Modes.sample_rate, 0, &g_cstate);
    return g_conv ? 0 : -1;
}

/* Converter only: iq -> mag, as one call of the reference iq_convert_fn. */
int ref_convert(int format, const void *iq, uint16_t *mag, unsigned n, double *mean_level, double *mean_power) {
    static int conv_format = -1;
    static iq_convert_fn fn;
    static struct converter_state *st;
    if (conv_format != format) {
        if (g_conv && !fn) { fn = g_conv; st = g_cstate; }
        else {
            Modes.sdr_type = SDR_NONE;
            fn = init_converter((input_format_t) format, 2400000.0, 0, &st);
        }
        conv_format = format;
    }
    if (!fn) return -1;
    fn((void *) iq, mag, n, st, mean_level, mean_power);
    return 0;
}

/* Whole path on an in-memory capture.  mag_dump (optional) receives the delayed
 * magnitude stream: mag_dump[0..326) = 0, mag_dump[326+i] = magnitude of sample i.
 * per-buffer mean_level/mean_power go to ml/mp (optional, nbuffers entries). */
int ref_demod_run(int format, int nfix, int fixdf, int thr,
                  const uint8_t *iq, uint64_t nsamples,
                  struct oracle_msg **out, uint64_t *nout, struct oracle_stats *st,
                  uint16_t *mag_dump, double *ml, double *mp) {
    if (ref_configure(format, nfix, fixdf, thr) < 0)
        return -1;
    const unsigned B = Modes.sdr_buf_samples, TR = Modes.trailing_samples;
    const unsigned bps = (format == ORACLE_FMT_UC8) ? 2 : 4;
    struct mag_buf bufs[2];
    memset(bufs, 0, sizeof(bufs));
    for (int i = 0; i < 2; i++)
        bufs[i].data = calloc(B + TR, sizeof(uint16_t));
    if (mag_dump) memset(mag_dump, 0, TR * sizeof(uint16_t));

    memset(st, 0, sizeof(*st));
    int64_t next_flip = 0;
    uint64_t sampleCounter = 0;
    uint64_t k = 0;
    int eof = 0;
    if (g_flip_before) {                        /* the decode thread found no buffer waiting on its first pass: backgroundTasks
                                                   (and the first filter flip) ran before buffer 0, readsb.c:857-902 */
        icaoFilterExpire();
        next_flip = Modes.synthetic_now + MODES_ICAO_FILTER_TTL;
        st->nflips++;
    }
    while (!eof) {
        struct mag_buf *outbuf = &bufs[k & 1], *lastbuf = &bufs[(k + 1) & 1];
        uint64_t remain = nsamples - sampleCounter;
        unsigned slen = remain >= B ? B : (unsigned) remain;
        if (slen < B) eof = 1;   /* short (possibly zero-length) read ends the file, sdr_ifile.c:223-237 */

        outbuf->sampleTimestamp = sampleCounter * 12e6 / Modes.sample_rate;
        if (k > 0 && lastbuf->length >= TR)
            memcpy(outbuf->data, lastbuf->data + lastbuf->length, TR * sizeof(uint16_t));
        else
            memset(outbuf->data, 0, TR * sizeof(uint16_t));
        outbuf->sysTimestamp = outbuf->sampleTimestamp / 12000U + Modes.startup_time;
        outbuf->sysMicroseconds = outbuf->sampleTimestamp / 12U + Modes.startup_time * 1000;
        outbuf->length = slen;
        outbuf->dropped = 0;

        double t0 = now_s();
        g_conv((void *) (iq + sampleCounter * bps), &outbuf->data[TR], slen, g_cstate, &outbuf->mean_level, &outbuf->mean_power);
        double t1 = now_s();
        if (mag_dump) memcpy(mag_dump + TR + sampleCounter, &outbuf->data[TR], slen * sizeof(uint16_t));
        if (ml) ml[k] = outbuf->mean_level;
        if (mp) mp[k] = outbuf->mean_power;
        sampleCounter += slen;

        demodulate2400(outbuf);
        if (g_mode_ac) demodulate2400AC(outbuf);
        double t2 = now_s();
        st->t_convert_s += t1 - t0;
        st->t_demod_s += t2 - t1;
        Modes.stats_current.samples_processed += outbuf->length;
        Modes.stats_current.samples_lost += B - outbuf->length;

        int64_t now = Modes.synthetic_now;      /* mstime() for ifile input, util.c:58-60 */
        if (now >= next_flip) {                 /* readsb.c:1227-1231 */
            icaoFilterExpire();
            next_flip = now + MODES_ICAO_FILTER_TTL;
            st->nflips++;
        }
        k++;
    }
    struct stats *s = &Modes.stats_current;
    st->demod_preambles = s->demod_preambles;
    st->demod_rejected_bad = s->demod_rejected_bad;
    st->demod_rejected_unknown_icao = s->demod_rejected_unknown_icao;
    for (int i = 0; i < 3; i++) st->demod_accepted[i] = s->demod_accepted[i];
    for (int i = 0; i < 5; i++) { st->demod_preamblePhase[i] = s->demod_preamblePhase[i]; st->demod_bestPhase[i] = s->demod_bestPhase[i]; }
    st->strong_signal_count = s->strong_signal_count;
    st->signal_power_count = s->signal_power_count;
    st->noise_power_count = s->noise_power_count;
    st->samples_processed = s->samples_processed;
    st->samples_lost = s->samples_lost;
    st->nbuffers = k;
    st->signal_power_sum = s->signal_power_sum;
    st->noise_power_sum = s->noise_power_sum;
    st->peak_signal_power = s->peak_signal_power;
    st->demod_modeac = s->demod_modeac;
    *out = g_out; *nout = g_nout;
    for (int i = 0; i < 2; i++) free(bufs[i].data);
    return 0;
}

/* CRC KATs (the reference's only pinned numbers for crc.c are the table sizes printed by
 * `crctests`): expose checksum + diagnose so tests can compare tables. */
/* ---- field decode: decodeModesMessage / decodeModeAMessage on one frame, struct modesMessage -> oracle_fields ---- */

static int g_fields_ready;

/* modesInit's order for what the decode needs: CRC tables, ICAO filter, the Mode A -> Mode C table. */
int ref_fields_init(int nfix) {
    if (g_fields_ready) return 0;
    if (!g_configured) {
        if (ref_configure(ORACLE_FMT_UC8, nfix, 1, 58) != 0) return -1;
    }
    modeACInit();
    g_fields_ready = 1;
    return 0;
}

static void fields_from_mm(const struct modesMessage *mm, struct oracle_fields *f) {
    memset(f, 0, sizeof(*f));
    f->addr = mm->addr; f->AA = mm->AA;
    f->flags = (mm->baro_alt_valid ? ORACLE_F_BARO_ALT_VALID : 0) | (mm->geom_alt_valid ? ORACLE_F_GEOM_ALT_VALID : 0)
        | (mm->heading_valid ? ORACLE_F_HEADING_VALID : 0) | (mm->gs_valid ? ORACLE_F_GS_VALID : 0)
        | (mm->ias_valid ? ORACLE_F_IAS_VALID : 0) | (mm->tas_valid ? ORACLE_F_TAS_VALID : 0)
        | (mm->baro_rate_valid ? ORACLE_F_BARO_RATE_VALID : 0) | (mm->geom_rate_valid ? ORACLE_F_GEOM_RATE_VALID : 0)
        | (mm->squawk_valid ? ORACLE_F_SQUAWK_VALID : 0) | (mm->callsign_valid ? ORACLE_F_CALLSIGN_VALID : 0)
        | (mm->cpr_valid ? ORACLE_F_CPR_VALID : 0) | (mm->cpr_odd ? ORACLE_F_CPR_ODD : 0)
        | (mm->category_valid ? ORACLE_F_CATEGORY_VALID : 0) | (mm->geom_delta_valid ? ORACLE_F_GEOM_DELTA_VALID : 0)
        | (mm->spi_valid ? ORACLE_F_SPI_VALID : 0) | (mm->spi ? ORACLE_F_SPI : 0)
        | (mm->alert_valid ? ORACLE_F_ALERT_VALID : 0) | (mm->alert ? ORACLE_F_ALERT : 0)
        | (mm->emergency_valid ? ORACLE_F_EMERGENCY_VALID : 0) | (mm->alt_q_bit ? ORACLE_F_ALT_Q_BIT : 0)
        | (mm->acas_ra_valid ? ORACLE_F_ACAS_RA_VALID : 0)
        | (mm->roll_valid ? ORACLE_F_ROLL_VALID : 0) | (mm->track_rate_valid ? ORACLE_F_TRACK_RATE_VALID : 0)
        | (mm->mach_valid ? ORACLE_F_MACH_VALID : 0) | (mm->wind_valid ? ORACLE_F_WIND_VALID : 0)
        | (mm->oat_valid ? ORACLE_F_OAT_VALID : 0) | (mm->static_pressure_valid ? ORACLE_F_STATIC_PRESSURE_VALID : 0)
        | (mm->turbulence_valid ? ORACLE_F_TURBULENCE_VALID : 0) | (mm->humidity_valid ? ORACLE_F_HUMIDITY_VALID : 0)
        | (mm->met_source_valid ? ORACLE_F_MET_SOURCE_VALID : 0);
    f->acc_flags = (mm->accuracy.nic_a_valid ? ORACLE_ACC_NIC_A_VALID : 0) | (mm->accuracy.nic_b_valid ? ORACLE_ACC_NIC_B_VALID : 0)
        | (mm->accuracy.nic_c_valid ? ORACLE_ACC_NIC_C_VALID : 0) | (mm->accuracy.nic_baro_valid ? ORACLE_ACC_NIC_BARO_VALID : 0)
        | (mm->accuracy.nac_p_valid ? ORACLE_ACC_NAC_P_VALID : 0) | (mm->accuracy.nac_v_valid ? ORACLE_ACC_NAC_V_VALID : 0)
        | (mm->accuracy.gva_valid ? ORACLE_ACC_GVA_VALID : 0) | (mm->accuracy.sda_valid ? ORACLE_ACC_SDA_VALID : 0)
        | (mm->accuracy.nic_a ? ORACLE_ACC_NIC_A : 0) | (mm->accuracy.nic_b ? ORACLE_ACC_NIC_B : 0)
        | (mm->accuracy.nic_c ? ORACLE_ACC_NIC_C : 0) | (mm->accuracy.nic_baro ? ORACLE_ACC_NIC_BARO : 0);
    f->nav_flags = (mm->nav.heading_valid ? ORACLE_NAV_HEADING_VALID : 0) | (mm->nav.fms_altitude_valid ? ORACLE_NAV_FMS_ALT_VALID : 0)
        | (mm->nav.mcp_altitude_valid ? ORACLE_NAV_MCP_ALT_VALID : 0) | (mm->nav.qnh_valid ? ORACLE_NAV_QNH_VALID : 0)
        | (mm->nav.modes_valid ? ORACLE_NAV_MODES_VALID : 0);
    f->msgtype = mm->msgtype; f->addrtype = mm->addrtype; f->source = mm->source; f->airground = mm->airground;
    f->metype = mm->metype; f->mesub = mm->mesub; f->CA = mm->CA; f->CC = mm->CC; f->CF = mm->CF;
    f->DR = mm->DR; f->FS = mm->FS; f->KE = mm->KE; f->ND = mm->ND; f->RI = mm->RI; f->SL = mm->SL; f->UM = mm->UM; f->VS = mm->VS;
    f->IID = mm->IID; f->category = mm->category; f->emergency = mm->emergency; f->cpr_type = mm->cpr_type;
    f->AC = mm->AC; f->ID = mm->ID; f->squawkHex = mm->squawkHex; f->squawkDec = mm->squawkDec;
    f->baro_alt = mm->baro_alt; f->geom_alt = mm->geom_alt; f->geom_delta = mm->geom_delta;
    f->baro_rate = mm->baro_rate; f->geom_rate = mm->geom_rate; f->ias = mm->ias; f->tas = mm->tas;
    f->heading = mm->heading; f->gs_v0 = mm->gs.v0; f->gs_v2 = mm->gs.v2; f->gs_selected = mm->gs.selected;
    f->cpr_lat = mm->cpr_lat; f->cpr_lon = mm->cpr_lon;
    memcpy(f->callsign, mm->callsign, 8);
    f->baro_alt_unit = mm->baro_alt_unit; f->geom_alt_unit = mm->geom_alt_unit; f->heading_type = mm->heading_type;
    f->sil_type = mm->accuracy.sil_type; f->nac_p = mm->accuracy.nac_p; f->nac_v = mm->accuracy.nac_v;
    f->sil = mm->accuracy.sil; f->gva = mm->accuracy.gva; f->sda = mm->accuracy.sda;
    f->op_version = mm->opstatus.version; f->op_hrd = mm->opstatus.hrd; f->op_tah = mm->opstatus.tah;
    f->op_flags = (mm->opstatus.valid ? ORACLE_OP_VALID : 0) | (mm->opstatus.om_acas_ra ? ORACLE_OP_OM_ACAS_RA : 0)
        | (mm->opstatus.om_ident ? ORACLE_OP_OM_IDENT : 0) | (mm->opstatus.om_atc ? ORACLE_OP_OM_ATC : 0)
        | (mm->opstatus.om_saf ? ORACLE_OP_OM_SAF : 0) | (mm->opstatus.cc_acas ? ORACLE_OP_CC_ACAS : 0)
        | (mm->opstatus.cc_cdti ? ORACLE_OP_CC_CDTI : 0) | (mm->opstatus.cc_1090_in ? ORACLE_OP_CC_1090_IN : 0)
        | (mm->opstatus.cc_arv ? ORACLE_OP_CC_ARV : 0) | (mm->opstatus.cc_ts ? ORACLE_OP_CC_TS : 0)
        | (mm->opstatus.cc_uat_in ? ORACLE_OP_CC_UAT_IN : 0) | (mm->opstatus.cc_poa ? ORACLE_OP_CC_POA : 0)
        | (mm->opstatus.cc_b2_low ? ORACLE_OP_CC_B2_LOW : 0) | (mm->opstatus.cc_lw_valid ? ORACLE_OP_CC_LW_VALID : 0);
    f->op_cc_lw = mm->opstatus.cc_lw; f->op_cc_antenna_offset = mm->opstatus.cc_antenna_offset; f->op_cc_tc = mm->opstatus.cc_tc;
    f->nav_heading_type = mm->nav.heading_type; f->nav_altitude_source = mm->nav.altitude_source; f->nav_modes = mm->nav.modes;
    f->nav_fms_altitude = mm->nav.fms_altitude; f->nav_mcp_altitude = mm->nav.mcp_altitude;
    f->nav_qnh = mm->nav.qnh; f->nav_heading = mm->nav.heading;
    f->roll = mm->roll; f->track_rate = mm->track_rate; f->mach = (float) mm->mach;
    if ((double) f->mach != mm->mach) { fprintf(stderr, "ref_harness: mach is not a float value\n"); exit(1); }
    f->oat = mm->oat; f->humidity = mm->humidity; f->wind_direction = mm->wind_direction;
    f->wind_speed = mm->wind_speed; f->static_pressure = mm->static_pressure;
    f->commb_format = mm->commb_format; f->met_source = mm->met_source; f->turbulence = mm->turbulence;
}

/* One frame (corrected bytes, as netUseMessage sees mm->msg): msgbits 56/112 -> decodeModesMessage, 16 -> the
 * Mode A/C reply decodeModeAMessage builds.  Address/Parity formats need their address in the ICAO filter to get
 * past the CRC stage (mode_s.c:477-482,572-581): it is added first, as it was when the frame was accepted in-stream.
 * Returns decodeResult (0 = decoded). */
int ref_decode_fields(const uint8_t *msg, int msgbits, struct oracle_fields *out) {
    static struct modesMessage mm;
    memset(&mm, 0, sizeof(mm));
    if (msgbits == 16) {
        decodeModeAMessage(&mm, (msg[0] << 8) | msg[1]);
        fields_from_mm(&mm, out);
        return 0;
    }
    memcpy(mm.msg, msg, 14);
    const int df = msg[0] >> 3;
    if (df != 11 && df != 17 && df != 18) icaoFilterAdd(modesChecksum((uint8_t *) msg, modesMessageLenByType(df)));
    const int rc = decodeModesMessage(&mm);
    fields_from_mm(&mm, out);
    return rc;
}

/* n frames of 14 bytes each; rc[i] = decodeResult.  The filter is flipped regularly so it does not fill up. */
void ref_decode_fields_batch(const uint8_t *msgs, const int32_t *msgbits, uint64_t n, struct oracle_fields *out, int32_t *rc) {
    for (uint64_t i = 0; i < n; ++i) {
        if ((i & 0xffff) == 0xffff) { icaoFilterExpire(); icaoFilterExpire(); }
        rc[i] = ref_decode_fields(msgs + 14 * i, msgbits[i], &out[i]);
    }
}

uint32_t ref_modesChecksum(const uint8_t *msg, int bits) { return modesChecksum((uint8_t *) msg, bits); }

/* returns number of error bits (0..2) or -1 when uncorrectable; bit positions in b0/b1 */
int ref_diagnose(uint32_t syndrome, int bits, int *b0, int *b1) {
    struct errorinfo *ei = modesChecksumDiagnose(syndrome, bits);
    if (!ei) return -1;
    *b0 = ei->errors > 0 ? ei->bit[0] : -1;
    *b1 = ei->errors > 1 ? ei->bit[1] : -1;
    return ei->errors;
}

void ref_crc_init(int nfix) { modesChecksumInit(nfix); }

#ifdef REF_HARNESS_MAIN
static void *read_file(const char *path, uint64_t *size) {
    FILE *f = fopen(path, "rb");
    if (!f) { perror(path); exit(1); }
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    void *p = malloc(sz ? sz : 1);
    if (fread(p, 1, sz, f) != (size_t) sz) { perror("fread"); exit(1); }
    fclose(f);
    *size = sz;
    return p;
}

/* ref_demod <UC8|SC16|SC16Q11> <nfix> <fixdf> <thr> <in.iq> <out.msgs> [out.stats] [out.mag]
 * out.msgs : array of struct oracle_msg;  out.stats : one struct oracle_stats;
 * out.mag  : u16 delayed magnitude stream (326 zeros + one magnitude per sample) */
int main(int argc, char **argv) {
    if (argc < 7) {
        fprintf(stderr, "usage: %s <UC8|SC16|SC16Q11> <nfix> <fixdf> <thr> <in.iq> <out.msgs> [out.stats] [out.mag]\n", argv[0]);
        return 2;
    }
    int format = !strcmp(argv[1], "UC8") ? ORACLE_FMT_UC8 : !strcmp(argv[1], "SC16") ? ORACLE_FMT_SC16 : ORACLE_FMT_SC16Q11;
    int nfix = atoi(argv[2]), fixdf = atoi(argv[3]), thr = atoi(argv[4]);
    uint64_t size;
    uint8_t *iq = read_file(argv[5], &size);
    uint64_t n = size / (format == ORACLE_FMT_UC8 ? 2 : 4);
    struct oracle_msg *out; uint64_t nout; struct oracle_stats st;
    uint16_t *mag = argc > 8 ? malloc((n + 326) * sizeof(uint16_t)) : NULL;
    if (getenv("ORACLE_MODE_AC")) ref_set_mode_ac(1);
    if (getenv("ORACLE_FLIP_BEFORE")) ref_set_flip_before(1);
    if (ref_demod_run(format, nfix, fixdf, thr, iq, n, &out, &nout, &st, mag, NULL, NULL) < 0)
        return 1;
    FILE *f = fopen(argv[6], "wb");
    if (!f) { perror(argv[6]); return 1; }
    fwrite(out, sizeof(*out), nout, f);
    fclose(f);
    if (argc > 7) { f = fopen(argv[7], "wb"); fwrite(&st, sizeof(st), 1, f); fclose(f); }
    if (argc > 8) { f = fopen(argv[8], "wb"); fwrite(mag, sizeof(uint16_t), n + 326, f); fclose(f); }
    fprintf(stderr, "ref_demod: %llu samples, %llu msgs, convert %.3f s, demod %.3f s\n",
            (unsigned long long) n, (unsigned long long) nout, st.t_convert_s, st.t_demod_s);
    return 0;
}
#endif
