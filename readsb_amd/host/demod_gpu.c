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
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>

int gpu_demod_open(struct gpu_demod *g, const struct mgpu_config *cfg, gpu_message_sink sink, void *user) {
    memset(g, 0, sizeof(*g));
    g->sink = sink;
    g->user = user;
    g->mode_ac = cfg->mode_ac != 0;
    int rc = mgpu_create(cfg, &g->ctx);
    if (rc != MGPU_OK) {
        fprintf(stderr, "gpu demodulator: %s\n", mgpu_strerror(rc));   /* loud: there is no CPU fallback */
        return rc;
    }
    g->scratch_cap = 65536;
    g->scratch = malloc(g->scratch_cap * sizeof(*g->scratch));
    return g->scratch ? MGPU_OK : MGPU_E_NOMEM;
}

/* The two chunk buffers of the file reader: allocated and page-locked once per demodulator (registration pins every page
 * and is serialised inside the driver — tens of ms per buffer, which must not sit in a stream's running time). */
int gpu_demod_reserve_input(struct gpu_demod *g, size_t bytes) {
    if (g->readbuf_bytes >= bytes) return MGPU_OK;
    for (int k = 0; k < 2; ++k) {
        if (g->readbuf[k]) {
            if (g->readbuf_pinned[k]) mgpu_host_unregister(g->ctx, g->readbuf[k]);
            free(g->readbuf[k]);
        }
        g->readbuf[k] = NULL;
        g->readbuf_pinned[k] = 0;
    }
    g->readbuf_bytes = 0;
    for (int k = 0; k < 2; ++k) {
        if (posix_memalign((void **) &g->readbuf[k], 4096, bytes) != 0) { g->readbuf[k] = NULL; return MGPU_E_NOMEM; }
        memset(g->readbuf[k], 0, bytes);                                             /* fault the pages in here, not in the reader */
        /* page-locked, the chunked uploads of mgpu_feed_iq run at PCIe speed beside the kernels (optional: ignore failure) */
        g->readbuf_pinned[k] = mgpu_host_register(g->ctx, g->readbuf[k], bytes) == MGPU_OK;
    }
    g->readbuf_bytes = bytes;
    return MGPU_OK;
}

void gpu_demod_close(struct gpu_demod *g) {
    for (int k = 0; k < 2; ++k) {
        if (g->readbuf[k] && g->readbuf_pinned[k] && g->ctx) mgpu_host_unregister(g->ctx, g->readbuf[k]);
        free(g->readbuf[k]);
    }
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
    /* with --modeac the decode thread runs demodulate2400AC(buf) right after demodulate2400(buf) (readsb.c:871-874) */
    /* `dropped` of the library entry = "Modes.stats_15min.samples_dropped != 0" (demod_2400.c:335-338), not this buffer's
     * own count: the host adds mag->dropped to its statistics (readsb.c:884-887) and the 15-minute window holds it for 15
     * minutes.  Stand-alone there is no struct stats: the window is kept here, on the buffers' own clock.
     * AN APPROXIMATION of the reference's timing, not its restatement: readsb's stats_15min is rebuilt when the periodic statistics
     * rotation folds the current one-minute bucket in (stats.c), so there the raised threshold starts at the first rotation AFTER the
     * drop and ends on a bucket boundary 15 buckets later; here it starts with the dropping buffer itself and ends exactly 15 minutes
     * on.  The link-time drop-in (readsb_tree/demod_gpu_wrap.c) reads the host's own Modes.stats_15min and is exact. */
    if (mag->dropped) { g->dropped_seen = 1; g->dropped_until_ms = mag->sysTimestamp + 15 * 60 * 1000; }
    const uint32_t dropped15 = g->dropped_seen && mag->sysTimestamp < g->dropped_until_ms;
    int rc = g->mode_ac ? mgpu_demod_mag_buf_ac(g->ctx, mag->data, mag->length, mag->sampleTimestamp, mag->sysTimestamp,
                                                mag->mean_level, mag->mean_power, dropped15)
                        : mgpu_demod_mag_buf(g->ctx, mag->data, mag->length, mag->sampleTimestamp, mag->sysTimestamp,
                                             mag->mean_power, dropped15);
    if (rc != MGPU_OK) {
        fprintf(stderr, "demodulate2400_gpu: %s (%s)\n", mgpu_strerror(rc), mgpu_last_error(g->ctx));
        abort();   /* the reference's demodulate2400 cannot fail; silently dropping a buffer would be worse */
    }
    deliver(g);
}

int gpu_ifile_run(struct gpu_demod *g, int fd, input_format_t format, unsigned chunk_buffers) {
    return gpu_ifile_run_until(g, fd, format, chunk_buffers, NULL, NULL);
}

/* ---- chunk reader: the next chunk is read while the current one is demodulated; a regular file is read by several
 * pread()s at once (one thread copies out of the page cache at a few GB/s, the GPU takes tens of GB/s) ---------------- */
enum { READ_SLICES = 8 };

struct slice_job { int fd; uint8_t *dst; size_t len; off_t off; size_t got; };

static void *slice_main(void *arg) {
    struct slice_job *j = arg;
    while (j->got < j->len) {
        ssize_t r = pread(j->fd, j->dst + j->got, j->len - j->got, j->off + (off_t) j->got);
        if (r < 0 && errno == EINTR) continue;
        if (r <= 0) break;
        j->got += (size_t) r;
    }
    return NULL;
}

struct chunk_reader {
    int fd, seekable, eof, stop;
    off_t pos, size;
    size_t want;                 /* bytes per chunk */
    uint8_t *buf[2];
    size_t have[2];
    int full[2];                 /* buffer holds a chunk the consumer has not released yet */
    pthread_mutex_t mu;
    pthread_cond_t cv;
};

static size_t read_chunk(struct chunk_reader *r, uint8_t *dst) {
    if (r->seekable) {
        const off_t left = r->size - r->pos;
        const size_t len = left <= 0 ? 0 : (size_t) left < r->want ? (size_t) left : r->want;
        struct slice_job job[READ_SLICES];
        pthread_t th[READ_SLICES];
        int started[READ_SLICES] = {0};
        const size_t per = (len / READ_SLICES + 4095) & ~(size_t) 4095;
        int n = 0;
        for (size_t at = 0; at < len && n < READ_SLICES; ++n) {
            const size_t l = (n == READ_SLICES - 1 || per == 0 || len - at < per) ? len - at : per;
            job[n] = (struct slice_job){r->fd, dst + at, l, r->pos + (off_t) at, 0};
            at += l;
        }
        for (int k = 1; k < n; ++k) started[k] = pthread_create(&th[k], NULL, slice_main, &job[k]) == 0;
        size_t got = 0;
        for (int k = 0; k < n; ++k) {
            if (k == 0 || !started[k]) slice_main(&job[k]); else pthread_join(th[k], NULL);
        }
        for (int k = 0; k < n; ++k) { got += job[k].got; if (job[k].got < job[k].len) break; }   /* a short slice ends the data */
        r->pos += (off_t) got;
        if (got < r->want) r->eof = 1;
        return got;
    }
    size_t have = 0;
    while (have < r->want) {                        /* sdr_ifile.c:221-235 */
        ssize_t n = read(r->fd, dst + have, r->want - have);
        if (n < 0 && errno == EINTR) continue;
        if (n <= 0) { r->eof = 1; break; }
        have += (size_t) n;
    }
    return have;
}

static void *reader_main(void *arg) {
    struct chunk_reader *r = arg;
    for (int k = 0;; k ^= 1) {
        pthread_mutex_lock(&r->mu);
        while (r->full[k] && !r->stop) pthread_cond_wait(&r->cv, &r->mu);
        const int stop = r->stop;
        pthread_mutex_unlock(&r->mu);
        if (stop) break;
        const size_t have = read_chunk(r, r->buf[k]);
        pthread_mutex_lock(&r->mu);
        r->have[k] = have;
        r->full[k] = 1;
        pthread_cond_broadcast(&r->cv);
        pthread_mutex_unlock(&r->mu);
        if (r->eof) break;                          /* the chunk just published (possibly empty) is the last one */
    }
    return NULL;
}

int gpu_ifile_run_until(struct gpu_demod *g, int fd, input_format_t format, unsigned chunk_buffers, const volatile int *stop, uint64_t *samples) {
    const size_t bps = format == INPUT_UC8 ? 2 : 4;
    const size_t buf_samples = 131072;
    const size_t chunk = (size_t) chunk_buffers * buf_samples;
    struct chunk_reader r;
    memset(&r, 0, sizeof(r));
    r.fd = fd;
    r.want = chunk * bps;
    struct stat st;
    if (fstat(fd, &st) == 0 && S_ISREG(st.st_mode)) {
        r.seekable = 1;
        r.size = st.st_size;
        r.pos = lseek(fd, 0, SEEK_CUR);
        if (r.pos < 0) r.pos = 0;
    }
    if (gpu_demod_reserve_input(g, r.want) != MGPU_OK) return MGPU_E_NOMEM;
    r.buf[0] = g->readbuf[0];
    r.buf[1] = g->readbuf[1];
    pthread_mutex_init(&r.mu, NULL);
    pthread_cond_init(&r.cv, NULL);
    pthread_t reader;
    const int threaded = pthread_create(&reader, NULL, reader_main, &r) == 0;
    int rc = MGPU_OK, last = 0;
    for (int k = 0; !last && !(stop && *stop); k ^= 1) {     /* while (!Modes.exit && !eof), sdr_ifile.c:197 */
        size_t have;
        if (threaded) {
            pthread_mutex_lock(&r.mu);
            while (!r.full[k]) pthread_cond_wait(&r.cv, &r.mu);
            have = r.have[k];
            pthread_mutex_unlock(&r.mu);
        } else {
            have = read_chunk(&r, r.buf[k]);
        }
        last = have < r.want;
        const uint64_t nsamples = have / bps;
        if (nsamples) {
            rc = mgpu_feed_iq(g->ctx, r.buf[k], nsamples);
            if (rc != MGPU_OK) break;
            if (samples) *samples += nsamples;
            deliver(g);
        }
        if (threaded) {
            pthread_mutex_lock(&r.mu);
            r.full[k] = 0;
            pthread_cond_broadcast(&r.cv);
            pthread_mutex_unlock(&r.mu);
        }
    }
    if (threaded) {
        pthread_mutex_lock(&r.mu);
        r.stop = 1;
        r.full[0] = r.full[1] = 0;
        pthread_cond_broadcast(&r.cv);
        pthread_mutex_unlock(&r.mu);
        pthread_join(reader, NULL);
    }
    if (rc == MGPU_OK) rc = mgpu_finish(g->ctx);    /* zero-length EOF buffer on exact multiples */
    if (rc == MGPU_OK) mgpu_collect(g->ctx, g->scratch, 0, NULL, &g->counters);
    pthread_mutex_destroy(&r.mu);
    pthread_cond_destroy(&r.cv);
    return rc;
}
