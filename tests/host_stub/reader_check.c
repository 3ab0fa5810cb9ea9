/* reader_check.c — CPU test of the host adapter's file reader (readsb_amd/host/demod_gpu.c: read-ahead thread, parallel
 * pread slices, EOF handling) against a stand-in for libmodes_gpu.so: the mgpu_* entry points below only record what they
 * are fed.  Built and run by tests/test_host_reader.py; no GPU involved, nothing of the product is replaced at run time.
 *
 *   reader_check <file|-> <UC8|SC16> <chunk_buffers> <out>     writes the bytes the adapter fed, in order, to <out>
 *   prints: feeds=<n> samples=<n> finish=<n> registered=<n> unregistered=<n> */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h>
#include <fcntl.h>
#include <unistd.h>

#include "../../readsb_amd/host/readsb_gpu_host.h"

struct mgpu_ctx { int format; uint64_t max_samples; };
static FILE *g_out;
static uint64_t g_feeds, g_samples, g_finish, g_reg, g_unreg, g_bad;
static void *g_regptr[8];

#undef mgpu_config_defaults   /* (the header's macro: hosts call mgpu_config_defaults_abi) */
static void config_defaults_(struct mgpu_config *cfg) { memset(cfg, 0, sizeof(*cfg)); cfg->abi_version = MGPU_ABI_VERSION; cfg->buf_samples = 131072; cfg->trailing_samples = 326; }
void mgpu_config_defaults(struct mgpu_config *cfg) { config_defaults_(cfg); }
void mgpu_config_defaults_abi(struct mgpu_config *cfg, uint32_t struct_bytes, uint32_t abi_version) { (void) struct_bytes; config_defaults_(cfg); cfg->abi_version = abi_version; }
int mgpu_create(const struct mgpu_config *cfg, mgpu_ctx **out) {
    *out = calloc(1, sizeof(**out)); (*out)->format = cfg->format; (*out)->max_samples = cfg->max_samples; return MGPU_OK;
}
void mgpu_destroy(mgpu_ctx *c) { free(c); }
const char *mgpu_strerror(int rc) { (void) rc; return "stub"; }
const char *mgpu_last_error(mgpu_ctx *c) { (void) c; return ""; }
int mgpu_host_register(mgpu_ctx *c, void *p, uint64_t bytes) { (void) c; (void) bytes; if (g_reg < 8) g_regptr[g_reg] = p; ++g_reg; return MGPU_OK; }
int mgpu_host_unregister(mgpu_ctx *c, void *p) { (void) c; (void) p; ++g_unreg; return MGPU_OK; }
int mgpu_feed_iq(mgpu_ctx *c, const void *iq, uint64_t nsamples) {
    const size_t bps = c->format == 0 ? 2 : 4;
    if (nsamples > c->max_samples) ++g_bad;                  /* a feed may not exceed cfg.max_samples */
    int known = 0;
    for (int k = 0; k < 8; ++k) known |= g_regptr[k] == iq;  /* fed straight from one of the page-locked buffers */
    if (!known) ++g_bad;
    fwrite(iq, bps, nsamples, g_out);
    ++g_feeds; g_samples += nsamples;
    return MGPU_OK;
}
int mgpu_finish(mgpu_ctx *c) { (void) c; ++g_finish; return MGPU_OK; }
int mgpu_collect(mgpu_ctx *c, struct mgpu_msg *out, uint64_t cap, uint64_t *n, struct mgpu_counters *k) {
    (void) c; (void) out; (void) cap; if (n) *n = 0; if (k) memset(k, 0, sizeof(*k)); return MGPU_OK;
}
int mgpu_demod_mag_buf(mgpu_ctx *c, const uint16_t *d, uint32_t l, int64_t a, int64_t b, double p, uint32_t dr) {
    (void) c; (void) d; (void) l; (void) a; (void) b; (void) p; (void) dr; return MGPU_OK;
}
int mgpu_demod_mag_buf_ac(mgpu_ctx *c, const uint16_t *d, uint32_t l, int64_t a, int64_t b, double ml, double p, uint32_t dr) {
    (void) c; (void) d; (void) l; (void) a; (void) b; (void) ml; (void) p; (void) dr; return MGPU_OK;
}

int main(int argc, char **argv) {
    if (argc != 5) return 2;
    const input_format_t fmt = !strcasecmp(argv[2], "UC8") ? INPUT_UC8 : INPUT_SC16;
    const unsigned chunk = (unsigned) atoi(argv[3]);
    const int fd = !strcmp(argv[1], "-") ? STDIN_FILENO : open(argv[1], O_RDONLY);
    if (fd < 0) return 3;
    g_out = fopen(argv[4], "wb");
    if (!g_out) return 4;
    struct mgpu_config cfg;
    mgpu_config_defaults(&cfg);
    cfg.format = (int) fmt;
    cfg.max_samples = (uint64_t) chunk * 131072;
    struct gpu_demod g;
    if (gpu_demod_open(&g, &cfg, NULL, NULL) != MGPU_OK) return 5;
    uint64_t samples = 0;
    const int rc = gpu_ifile_run_until(&g, fd, fmt, chunk, NULL, &samples);
    gpu_demod_close(&g);
    fclose(g_out);
    printf("rc=%d feeds=%llu samples=%llu counted=%llu finish=%llu registered=%llu unregistered=%llu bad=%llu\n", rc, (unsigned long long) g_feeds,
           (unsigned long long) g_samples, (unsigned long long) samples, (unsigned long long) g_finish, (unsigned long long) g_reg,
           (unsigned long long) g_unreg, (unsigned long long) g_bad);
    return rc == MGPU_OK ? 0 : 1;
}
