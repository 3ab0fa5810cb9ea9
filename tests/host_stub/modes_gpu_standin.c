/* modes_gpu_standin.c — TEST INFRASTRUCTURE: a CPU stand-in with libmodes_gpu.so's C ABI (the struct mag_buf entry only),
 * implemented by the restated oracle, so that the link-time drop-in of the whole reference program
 * (readsb_amd/host/readsb_tree/demod_gpu_wrap.c, `make -C oracle full_standin`) can be exercised without a GPU:
 * what is under test is the ADAPTER inside the reference program, not the demodulator.  Never shipped, never linked
 * into the product; the product library has no CPU path. */
#include <stdlib.h>
#include <string.h>

#include "../../include/modes_gpu.h"
#include "../../oracle/modes_oracle.h"

struct mgpu_ctx { struct mgpu_config cfg; struct oracle_msg *pending; uint64_t npending, next; struct oracle_stats st; };

void mgpu_config_defaults(struct mgpu_config *cfg) {
    memset(cfg, 0, sizeof(*cfg));
    cfg->nfix_crc = 1; cfg->fixDF = 1; cfg->preamble_threshold = 58; cfg->buf_samples = 131072; cfg->trailing_samples = 326;
}
int mgpu_create(const struct mgpu_config *cfg, mgpu_ctx **out) {
    mgpu_ctx *c = calloc(1, sizeof(*c));
    c->cfg = *cfg;
    const struct modes_oracle_cfg oc = {0, cfg->nfix_crc, cfg->fixDF, cfg->preamble_threshold};
    modes_oracle_set_mode_ac((int) cfg->mode_ac);
    modes_oracle_stream_begin(&oc, cfg->startup_time_ms);
    *out = c;
    return MGPU_OK;
}
void mgpu_destroy(mgpu_ctx *c) { if (c) { modes_oracle_free(c->pending); free(c); } }
const char *mgpu_strerror(int rc) { (void) rc; return "stand-in"; }
const char *mgpu_last_error(mgpu_ctx *c) { (void) c; return ""; }

static int run(mgpu_ctx *c, const uint16_t *data, uint32_t length, int64_t st, int64_t sys, double ml, double mp) {
    if (c->next < c->npending) return MGPU_E_INVAL;          /* collect first */
    modes_oracle_free(c->pending);
    modes_oracle_stream_mag_buf(data, length, st, sys, ml, mp);
    c->npending = modes_oracle_stream_take(&c->pending, &c->st);
    c->next = 0;
    return MGPU_OK;
}
int mgpu_demod_mag_buf(mgpu_ctx *c, const uint16_t *data, uint32_t length, int64_t st, int64_t sys, double mp, uint32_t dropped) {
    (void) dropped;
    if (c->cfg.mode_ac) return MGPU_E_INVAL;
    return run(c, data, length, st, sys, 0.0, mp);
}
int mgpu_demod_mag_buf_ac(mgpu_ctx *c, const uint16_t *data, uint32_t length, int64_t st, int64_t sys, double ml, double mp, uint32_t dropped) {
    (void) dropped;
    return run(c, data, length, st, sys, ml, mp);
}
int mgpu_collect(mgpu_ctx *c, struct mgpu_msg *out, uint64_t cap, uint64_t *n, struct mgpu_counters *k) {
    uint64_t m = 0;
    while (m < cap && c->next < c->npending) {
        const struct oracle_msg *o = &c->pending[c->next++];
        struct mgpu_msg *g = &out[m++];
        memset(g, 0, sizeof(*g));
        g->timestamp = o->timestamp;
        g->sysTimestamp = o->sys_rel_ms + ORACLE_STARTUP_MS;
        g->sig_len = (uint16_t) (o->msgbits == 112 ? 268 : 134);
        /* signalLevel = sumsq / 65535 / 65535 / len exactly (demod_2400.c:447-448): recover the integer sum */
        g->sig_sumsq = (uint64_t) (o->signalLevel * g->sig_len * 65535.0 * 65535.0 + 0.5);
        g->score = (int16_t) o->score;
        g->correctedbits = (uint8_t) o->correctedbits;
        g->msgtype = (uint8_t) o->msgtype;
        g->msgbits = (uint8_t) o->msgbits;
        g->addr = o->addr;
        memcpy(g->msg, o->msg, 14);
        memcpy(g->raw, o->raw, 14);
    }
    if (n) *n = m;
    if (k) {
        memset(k, 0, sizeof(*k));
        const struct oracle_stats *s = &c->st;
        k->demod_preambles = s->demod_preambles; k->demod_rejected_bad = s->demod_rejected_bad;
        k->demod_rejected_unknown_icao = s->demod_rejected_unknown_icao;
        for (int i = 0; i < 3; ++i) k->demod_accepted[i] = s->demod_accepted[i];
        for (int i = 0; i < 5; ++i) { k->demod_preamblePhase[i] = s->demod_preamblePhase[i]; k->demod_bestPhase[i] = s->demod_bestPhase[i]; }
        k->strong_signal_count = s->strong_signal_count; k->signal_power_count = s->signal_power_count;
        k->noise_power_count = s->noise_power_count; k->samples_processed = s->samples_processed; k->samples_lost = s->samples_lost;
        k->nbuffers = s->nbuffers; k->nflips = s->nflips; k->signal_power_sum = s->signal_power_sum;
        k->noise_power_sum = s->noise_power_sum; k->peak_signal_power = s->peak_signal_power; k->demod_modeac = s->demod_modeac;
    }
    return MGPU_OK;
}
