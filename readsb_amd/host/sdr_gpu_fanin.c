/* sdr_gpu_fanin.c — many sample files in, one GPU demodulator context per stream.
 *
 * Shaped like a row of the reference's sdr_handlers table (sdr.c:94-122) and like its ifile handler
 * (sdr_ifile.c: ifileInitConfig :95, ifileHandleOption :103, ifileOpen :116, ifileRun :169, ifileClose :272), with the
 * per-stream state in a struct instead of the reference's single global `ifile`.  What a stream does is
 * gpu_ifile_run() (demod_gpu.c): read a chunk into a page-locked buffer, mgpu_feed_iq(), deliver the messages.
 */
#include "readsb_gpu_host.h"

#include <errno.h>
#include <fcntl.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h>
#include <unistd.h>

void gpuFaninInitConfig(struct gpu_fanin *f) {
    memset(f, 0, sizeof(*f));
    mgpu_config_defaults(&f->cfg);
    f->next_format = INPUT_UC8;            /* ifileInitConfig: input_format = INPUT_UC8, sdr_ifile.c:97 */
    f->chunk_buffers = 256;
}

static int add_stream(struct gpu_fanin *f, const char *path) {
    if (f->nstreams == f->cap) {
        const unsigned cap = f->cap ? f->cap * 2 : 8;
        struct gpu_fanin_stream *s = realloc(f->streams, cap * sizeof(*s));
        if (!s) return MGPU_E_NOMEM;
        f->streams = s;
        f->cap = cap;
    }
    struct gpu_fanin_stream *s = &f->streams[f->nstreams];
    memset(s, 0, sizeof(*s));
    s->path = strdup(path);
    if (!s->path) return MGPU_E_NOMEM;
    s->fd = -1;
    s->format = f->next_format;
    s->index = f->nstreams++;
    return MGPU_OK;
}

int gpuFaninHandleOption(struct gpu_fanin *f, const char *opt, const char *arg) {
    if (!strcmp(opt, "--ifile")) {
        if (!arg) return 0;
        return add_stream(f, arg) == MGPU_OK ? 2 : 0;
    }
    if (!strcmp(opt, "--iformat")) {       /* ifileHandleOption, sdr_ifile.c:107-113 */
        if (!arg) return 0;
        if (!strcasecmp(arg, "uc8")) f->next_format = INPUT_UC8;
        else if (!strcasecmp(arg, "sc16")) f->next_format = INPUT_SC16;
        else if (!strcasecmp(arg, "sc16q11")) f->next_format = INPUT_SC16Q11;
        else { fprintf(stderr, "Input format '%s' not understood (supported values: UC8, SC16, SC16Q11)\n", arg); return 0; }
        return 2;
    }
    if (!strcmp(opt, "--gpu-devices")) { if (!arg) return 0; f->devices = atoi(arg); return 2; }
    if (!strcmp(opt, "--gpu-chunk-buffers")) { if (!arg) return 0; f->chunk_buffers = (unsigned) atoi(arg); return 2; }
    if (!strcmp(opt, "--fix")) { f->cfg.nfix_crc = 1; return 1; }
    if (!strcmp(opt, "--no-fix")) { f->cfg.nfix_crc = 0; return 1; }
    if (!strcmp(opt, "--aggressive")) { f->cfg.nfix_crc = 2; return 1; }
    if (!strcmp(opt, "--no-fix-df")) { f->cfg.fixDF = 0; return 1; }
    if (!strcmp(opt, "--modeac")) { f->cfg.mode_ac = 1; return 1; }
    if (!strcmp(opt, "--preamble-threshold")) { if (!arg) return 0; f->cfg.preamble_threshold = atoi(arg); return 2; }
    if (!strcmp(opt, "--startup-time-ms")) { if (!arg) return 0; f->cfg.startup_time_ms = atoll(arg); return 2; }
    return 0;
}

static void stream_sink(const struct gpu_modes_message *mm, void *user) {
    struct gpu_fanin_stream *s = user;
    if (s->owner->sink) s->owner->sink(s->index, mm, s->owner->user);
}

int gpuFaninOpen(struct gpu_fanin *f, gpu_stream_sink sink, void *user) {
    if (f->nstreams == 0) {
        fprintf(stderr, "SDR type 'ifile' requires an --ifile argument\n");      /* sdr_ifile.c:118 */
        return MGPU_E_INVAL;
    }
    f->sink = sink;
    f->user = user;
    int ndev = mgpu_device_count();
    if (ndev <= 0) {
        fprintf(stderr, "gpu fan-in: no HIP device (there is no CPU path)\n");
        return MGPU_E_NODEVICE;
    }
    if (f->devices > 0 && f->devices < ndev) ndev = f->devices;
    if (f->chunk_buffers == 0) f->chunk_buffers = 256;
    for (unsigned i = 0; i < f->nstreams; ++i) {
        struct gpu_fanin_stream *s = &f->streams[i];
        s->owner = f;
        s->fd = !strcmp(s->path, "-") ? STDIN_FILENO : open(s->path, O_RDONLY);      /* ifileOpen, sdr_ifile.c:122-130 */
        if (s->fd < 0) {
            fprintf(stderr, "ifile: could not open %s: %s\n", s->path, strerror(errno));
            return MGPU_E_INVAL;
        }
        struct mgpu_config cfg = f->cfg;
        cfg.device = (int) (i % (unsigned) ndev);
        /* how many of the streams land on this device: with more than one, every context keeps its host stages small */
        cfg.streams_on_device = (f->nstreams - (unsigned) cfg.device + (unsigned) ndev - 1) / (unsigned) ndev;
        cfg.format = (int) s->format;
        cfg.max_samples = (uint64_t) f->chunk_buffers * 131072;
        s->device = cfg.device;
        const int rc = gpu_demod_open(&s->demod, &cfg, stream_sink, s);
        if (rc != MGPU_OK) return rc;
        const size_t bps = s->format == INPUT_UC8 ? 2 : 4;
        if (gpu_demod_reserve_input(&s->demod, (size_t) f->chunk_buffers * 131072 * bps) != MGPU_OK) return MGPU_E_NOMEM;
    }
    return MGPU_OK;
}

static void *stream_main(void *arg) {
    struct gpu_fanin_stream *s = arg;
    s->rc = gpu_ifile_run_until(&s->demod, s->fd, s->format, s->owner->chunk_buffers, &s->owner->exit, &s->samples);
    return NULL;
}

int gpuFaninRun(struct gpu_fanin *f) {
    for (unsigned i = 0; i < f->nstreams; ++i) {
        struct gpu_fanin_stream *s = &f->streams[i];
        pthread_t t;
        if (pthread_create(&t, NULL, stream_main, s) != 0) { s->rc = MGPU_E_NOMEM; f->exit = 1; break; }
        s->thread = (unsigned long) t;
        s->started = 1;
    }
    int rc = MGPU_OK;
    for (unsigned i = 0; i < f->nstreams; ++i) {
        struct gpu_fanin_stream *s = &f->streams[i];
        if (s->started) pthread_join((pthread_t) s->thread, NULL);
        s->started = 0;
        if (rc == MGPU_OK && s->rc != MGPU_OK) rc = s->rc;
    }
    return rc;
}

void gpuFaninCancel(struct gpu_fanin *f) { f->exit = 1; }

void gpuFaninClose(struct gpu_fanin *f) {
    for (unsigned i = 0; i < f->nstreams; ++i) {
        struct gpu_fanin_stream *s = &f->streams[i];
        if (s->demod.ctx) gpu_demod_close(&s->demod);
        if (s->fd >= 0 && s->fd != STDIN_FILENO) close(s->fd);           /* ifileClose, sdr_ifile.c:272-283 */
        free(s->path);
    }
    free(f->streams);
    memset(f, 0, sizeof(*f));
}
