/* fanin_check.c — CPU test of the fan-in row (readsb_amd/host/sdr_gpu_fanin.c: option handling, one context per stream,
 * streams spread over the devices, one reader/feeder thread per stream) against a recording stand-in for libmodes_gpu.so.
 *   fanin_check <prefix> [fan-in options…]    every context writes what it is fed to <prefix>.<creation index>
 * prints one line per stream: stream=<k> device=<d> format=<f> samples=<n> ctx=<creation index> */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../readsb_amd/host/readsb_gpu_host.h"

struct mgpu_ctx { int format, device, index; uint64_t max_samples, samples; FILE *out; };
static const char *g_prefix;
static int g_nctx, g_finish;
static pthread_mutex_t g_mu = PTHREAD_MUTEX_INITIALIZER;

#undef mgpu_config_defaults   /* (the header's macro: hosts call mgpu_config_defaults_abi) */
static void config_defaults_(struct mgpu_config *cfg) { memset(cfg, 0, sizeof(*cfg)); cfg->abi_version = MGPU_ABI_VERSION; cfg->nfix_crc = 1; cfg->fixDF = 1; cfg->preamble_threshold = 58; }
void mgpu_config_defaults(struct mgpu_config *cfg) { config_defaults_(cfg); }
void mgpu_config_defaults_abi(struct mgpu_config *cfg, uint32_t struct_bytes, uint32_t abi_version) { (void) struct_bytes; config_defaults_(cfg); cfg->abi_version = abi_version; }
int mgpu_device_count(void) { return 3; }
int mgpu_create(const struct mgpu_config *cfg, mgpu_ctx **out) {
    mgpu_ctx *c = calloc(1, sizeof(*c));
    c->format = cfg->format; c->device = cfg->device; c->max_samples = cfg->max_samples;
    pthread_mutex_lock(&g_mu); c->index = g_nctx++; pthread_mutex_unlock(&g_mu);
    char name[4096];
    snprintf(name, sizeof(name), "%s.%d", g_prefix, c->index);
    c->out = fopen(name, "wb");
    *out = c;
    return c->out ? MGPU_OK : MGPU_E_INVAL;
}
void mgpu_destroy(mgpu_ctx *c) { if (c->out) fclose(c->out); free(c); }
const char *mgpu_strerror(int rc) { (void) rc; return "stub"; }
const char *mgpu_last_error(mgpu_ctx *c) { (void) c; return ""; }
int mgpu_host_register(mgpu_ctx *c, void *p, uint64_t bytes) { (void) c; (void) p; (void) bytes; return MGPU_OK; }
int mgpu_host_unregister(mgpu_ctx *c, void *p) { (void) c; (void) p; return MGPU_OK; }
int mgpu_feed_iq(mgpu_ctx *c, const void *iq, uint64_t nsamples) {
    if (nsamples > c->max_samples) return MGPU_E_INVAL;
    fwrite(iq, c->format == 0 ? 2 : 4, nsamples, c->out);
    c->samples += nsamples;
    return MGPU_OK;
}
int mgpu_finish(mgpu_ctx *c) { (void) c; pthread_mutex_lock(&g_mu); ++g_finish; pthread_mutex_unlock(&g_mu); return MGPU_OK; }
int mgpu_collect(mgpu_ctx *c, struct mgpu_msg *out, uint64_t cap, uint64_t *n, struct mgpu_counters *k) {
    (void) out; (void) cap; if (n) *n = 0; if (k) { memset(k, 0, sizeof(*k)); k->samples_processed = c->samples; } return MGPU_OK;
}
int mgpu_demod_mag_buf(mgpu_ctx *c, const uint16_t *d, uint32_t l, int64_t a, int64_t b, double p, uint32_t dr) {
    (void) c; (void) d; (void) l; (void) a; (void) b; (void) p; (void) dr; return MGPU_OK;
}
int mgpu_demod_mag_buf_ac(mgpu_ctx *c, const uint16_t *d, uint32_t l, int64_t a, int64_t b, double ml, double p, uint32_t dr) {
    (void) c; (void) d; (void) l; (void) a; (void) b; (void) ml; (void) p; (void) dr; return MGPU_OK;
}

int main(int argc, char **argv) {
    if (argc < 2) return 2;
    g_prefix = argv[1];
    struct gpu_fanin f;
    gpuFaninInitConfig(&f);
    for (int i = 2; i < argc; i++) {
        const int used = gpuFaninHandleOption(&f, argv[i], i + 1 < argc ? argv[i + 1] : NULL);
        if (!used) { fprintf(stderr, "not a fan-in option: %s\n", argv[i]); return 2; }
        i += used - 1;
    }
    int rc = gpuFaninOpen(&f, NULL, NULL);
    if (rc != MGPU_OK) { printf("open=%d\n", rc); gpuFaninClose(&f); return 1; }
    rc = gpuFaninRun(&f);
    printf("run=%d streams=%u finish=%d nfix=%d mode_ac=%u thr=%d\n", rc, f.nstreams, g_finish, f.cfg.nfix_crc, f.cfg.mode_ac, f.cfg.preamble_threshold);
    for (unsigned k = 0; k < f.nstreams; ++k)
        printf("stream=%u device=%d format=%d samples=%llu ctx=%d processed=%llu\n", k, f.streams[k].device, (int) f.streams[k].format,
               (unsigned long long) f.streams[k].samples, f.streams[k].demod.ctx->index,
               (unsigned long long) f.streams[k].demod.counters.samples_processed);
    gpuFaninClose(&f);
    return rc == MGPU_OK ? 0 : 1;
}
