/* demod_gpu.c — adapter between readsb's decode-thread interfaces and libmodes_gpu.so.
 *
 *   demodulate2400_gpu()  replaces  demodulate2400(struct mag_buf *)        demod_2400.c:264
 *   gpu_ifile_run()       replaces  ifileRun()'s read/convert/push loop      sdr_ifile.c:169-270
 *
 * Per accepted message the reference does: mm = netGetMM(); fill timestamp/sysTimestamp/score/
 * msg; decodeModesMessage(mm); signalLevel; netUseMessage(mm) (demod_2400.c:401-471).  The GPU
 * path returns exactly those fields (struct mgpu_msg), in the same order, so the loop below is
 * the whole host side.  See INTEGRATION.md for the variant compiled inside a readsb tree.
 */
#include "readsb_gpu_host.h"

#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

int gpu_demod_open(struct gpu_demod *g, const struct mgpu_config *cfg, gpu_message_sink sink, void *user) {
    memset(g, 0, sizeof(*g));
    g->sink = sink;
    g->user = user;
    int rc = mgpu_create(cfg, &g->ctx);
    if (rc != MGPU_OK) {
        fprintf(stderr, "gpu demodulator: %s\n", mgpu_strerror(rc));   /* loud: there is no CPU fallback */
        return rc;
    }
    g->scratch_cap = 65536;
    g->scratch = malloc(g->scratch_cap * sizeof(*g->scratch));
    return g->scratch ? MGPU_OK : MGPU_E_NOMEM;
}

void gpu_demod_close(struct gpu_demod *g) {
    if (g->ctx) mgpu_destroy(g->ctx);
    free(g->scratch);
    memset(g, 0, sizeof(*g));
}

/* netGetMM -> fill -> decodeModesMessage -> netUseMessage, for every pending message */
static void deliver(struct gpu_demod *g) {
    for (;;) {
        uint64_t n = 0;
        if (mgpu_collect(g->ctx, g->scratch, g->scratch_cap, &n, &g->counters) != MGPU_OK || n == 0)
            break;
        for (uint64_t i = 0; i < n; ++i) {
            const struct mgpu_msg *m = &g->scratch[i];
            struct gpu_modes_message mm;
            memset(&mm, 0, sizeof(mm));
            memcpy(mm.verbatim, m->raw, 14);
            memcpy(mm.msg, m->msg, 14);
            mm.timestamp = m->timestamp;
            mm.sysTimestamp = m->sysTimestamp;
            mm.score = m->score;
            mm.msgtype = m->msgtype;
            mm.msgbits = m->msgbits;
            mm.correctedbits = m->correctedbits;
            mm.addr = m->addr;
            mm.signalLevel = mgpu_msg_signal_level(m);
            if (g->sink) g->sink(&mm, g->user);
        }
    }
}

void demodulate2400_gpu(struct gpu_demod *g, struct mag_buf *mag) {
    int rc = mgpu_demod_mag_buf(g->ctx, mag->data, mag->length, mag->sampleTimestamp, mag->sysTimestamp,
                                mag->mean_power, mag->dropped);
    if (rc != MGPU_OK) {
        fprintf(stderr, "demodulate2400_gpu: %s (%s)\n", mgpu_strerror(rc), mgpu_last_error(g->ctx));
        abort();   /* the reference's demodulate2400 cannot fail; silently dropping a buffer would be worse */
    }
    deliver(g);
}

int gpu_ifile_run(struct gpu_demod *g, int fd, input_format_t format, unsigned chunk_buffers) {
    return gpu_ifile_run_until(g, fd, format, chunk_buffers, NULL, NULL);
}

int gpu_ifile_run_until(struct gpu_demod *g, int fd, input_format_t format, unsigned chunk_buffers, const volatile int *stop, uint64_t *samples) {
    const size_t bps = format == INPUT_UC8 ? 2 : 4;
    const size_t buf_samples = 131072;
    const size_t chunk = (size_t) chunk_buffers * buf_samples;
    uint8_t *readbuf = malloc(chunk * bps);
    if (!readbuf) return MGPU_E_NOMEM;
    /* page-locked, the chunked uploads of mgpu_feed_iq run at PCIe speed beside the kernels (optional: ignore failure) */
    const int pinned = mgpu_host_register(g->ctx, readbuf, chunk * bps) == MGPU_OK;
    int rc = MGPU_OK, eof = 0;
    while (!eof && !(stop && *stop)) {         /* while (!Modes.exit && !eof), sdr_ifile.c:197 */
        size_t have = 0, want = chunk * bps;
        while (have < want) {                       /* sdr_ifile.c:221-235 */
            ssize_t r = read(fd, readbuf + have, want - have);
            if (r < 0 && errno == EINTR) continue;
            if (r <= 0) { eof = 1; break; }
            have += (size_t) r;
        }
        const uint64_t nsamples = have / bps;
        if (nsamples) {
            rc = mgpu_feed_iq(g->ctx, readbuf, nsamples);
            if (rc != MGPU_OK) break;
            if (samples) *samples += nsamples;
            deliver(g);
        }
    }
    if (rc == MGPU_OK) rc = mgpu_finish(g->ctx);    /* zero-length EOF buffer on exact multiples */
    if (rc == MGPU_OK) mgpu_collect(g->ctx, g->scratch, 0, NULL, &g->counters);
    if (pinned) mgpu_host_unregister(g->ctx, readbuf);
    free(readbuf);
    return rc;
}
