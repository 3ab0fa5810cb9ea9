/* readsb_gpu_host.h — host-side C mirror of the reference interfaces on either side of the
 * GPU hot path, so the library drops into readsb's reader/decode threads.
 *
 * In a readsb tree this file is not needed: the adapter (demod_gpu.c) is compiled against
 * readsb.h and uses the real `struct mag_buf`, `struct modesMessage`, netGetMM(),
 * decodeModesMessage() and netUseMessage().  Stand-alone (this repo),
 * the same adapter is compiled against the minimal look-alikes below: identical field
 * names and meaning for everything the path touches, nothing else.
 */
#ifndef READSB_GPU_HOST_H
#define READSB_GPU_HOST_H

#include <stdbool.h>
#include <stdint.h>
#include "../../include/modes_gpu.h"

#ifdef __cplusplus
extern "C" {
#endif

/* input_format_t, convert.h:28-31 */
typedef enum { INPUT_UC8 = 0, INPUT_SC16, INPUT_SC16Q11 } input_format_t;

/* struct mag_buf, readsb.h:450-464 (same fields, same order) */
struct mag_buf {
    int64_t sampleTimestamp;  /* 12 MHz clock at the start of this block */
    double mean_level;
    double mean_power;
    uint32_t dropped;
    unsigned length;          /* valid samples after the overlap */
    int64_t sysTimestamp;
    int64_t sysMicroseconds;
    uint16_t *data;           /* trailing_samples of overlap, then `length` samples */
};

/* The slice of struct modesMessage (readsb.h:887-1144) demodulate2400 fills before/after
 * decodeModesMessage (demod_2400.c:401-471). */
struct gpu_modes_message {
    unsigned char msg[14];       /* mm->msg (corrected by decodeModesMessage) */
    unsigned char verbatim[14];  /* mm->verbatim: as received */
    double signalLevel;          /* mm->signalLevel */
    int64_t timestamp;           /* mm->timestamp */
    int64_t sysTimestamp;        /* mm->sysTimestamp */
    int msgtype, msgbits, score, correctedbits;
    uint32_t addr;
};

/* What demodulate2400 calls per accepted message in the reference: netGetMM + decodeModesMessage
 * + netUseMessage (net_io.c:5978-6006, mode_s.c:443).  Stand-alone we deliver the filled message
 * to a callback; in a readsb tree the adapter calls the real functions instead. */
typedef void (*gpu_message_sink)(const struct gpu_modes_message *mm, void *user);

struct gpu_demod {
    mgpu_ctx *ctx;
    gpu_message_sink sink;
    void *user;
    struct mgpu_msg *scratch;
    uint64_t scratch_cap;
    struct mgpu_counters counters;   /* mirrors Modes.stats_current's demod counters */
    int dropped_seen;                /* a buffer with dropped samples has passed ... */
    int64_t dropped_until_ms;        /* ... and counts, as in Modes.stats_15min, until this sysTimestamp */
    uint8_t *readbuf[2];             /* the file reader's two page-locked chunk buffers (gpu_demod_reserve_input) */
    size_t readbuf_bytes;
    int readbuf_pinned[2];
    int mode_ac;                     /* Modes.mode_ac: demodulate2400_gpu also runs the Mode A/C demodulator */
};

/* modesInit()'s hot-path part (readsb.c:285-310) + init_converter (sdr_ifile.c:156) */
int gpu_demod_open(struct gpu_demod *g, const struct mgpu_config *cfg, gpu_message_sink sink, void *user);
void gpu_demod_close(struct gpu_demod *g);
/* optional: allocate and page-lock the file reader's buffers now (else the first gpu_ifile_run does it) */
int gpu_demod_reserve_input(struct gpu_demod *g, size_t bytes);

/* void demodulate2400(struct mag_buf *mag) (demod_2400.h:38): same call shape, decode thread only */
void demodulate2400_gpu(struct gpu_demod *g, struct mag_buf *mag);

/* Bulk path for file input: ifileRun's read loop (sdr_ifile.c:169-270) collapsed into large feeds:
 * reads `fd` to EOF in chunks of `chunk_buffers` 131072-sample buffers, converts and demodulates
 * on the GPU, delivers messages in stream order.  Returns 0 or a negative MGPU_E_* code. */
int gpu_ifile_run(struct gpu_demod *g, int fd, input_format_t format, unsigned chunk_buffers);
/* the same with the reader loop's exit flag (Modes.exit, sdr_ifile.c:197) and a running sample count; both may be NULL */
int gpu_ifile_run_until(struct gpu_demod *g, int fd, input_format_t format, unsigned chunk_buffers, const volatile int *stop, uint64_t *samples);

/* ---- fan-in: many sample streams, one demodulator context each -------------------------------
 * The aggregator's input side (README.md:40-51: several receivers feeding one readsb): an sdr_handler-shaped row
 * (sdr.c:94-122: initConfig / handleOption / open / run / cancel / close) that takes any number of `--ifile`s.  Every
 * stream gets its own context — its own sample clock, ICAO filter and counters, exactly one readsb process's worth —
 * on GPU `stream index % devices`; each stream's reader thread reads into its own page-locked buffer and feeds its
 * context, so one stream's file reads and uploads run under the others' kernels.  Messages arrive per stream, in that
 * stream's order, on the stream's thread. */
typedef void (*gpu_stream_sink)(unsigned stream, const struct gpu_modes_message *mm, void *user);

struct gpu_fanin;
struct gpu_fanin_stream {
    char *path;
    int fd;
    input_format_t format;
    int device;
    struct gpu_demod demod;
    struct gpu_fanin *owner;
    unsigned index;
    int started;                 /* thread exists */
    int rc;                      /* MGPU_OK or the code that ended the stream */
    uint64_t samples;            /* samples demodulated */
    unsigned long thread;        /* pthread_t */
};

struct gpu_fanin {
    struct gpu_fanin_stream *streams;
    unsigned nstreams, cap;
    struct mgpu_config cfg;      /* options common to all streams (--fix, --modeac, threshold, …) */
    input_format_t next_format;  /* --iformat applies to the --ifile arguments that follow it */
    unsigned chunk_buffers;      /* 131072-sample buffers per feed */
    int devices;                 /* GPUs to spread the streams over; 0 = all visible */
    gpu_stream_sink sink;
    void *user;
    volatile int exit;           /* Modes.exit's role for the reader loops (sdr_ifile.c:197) */
};

void gpuFaninInitConfig(struct gpu_fanin *f);
/* `--ifile PATH` (repeatable), `--iformat UC8|SC16|SC16Q11`, `--gpu-devices N`, `--gpu-chunk-buffers N`,
 * `--fix` `--no-fix` `--aggressive` `--no-fix-df` `--modeac` `--preamble-threshold N` `--startup-time-ms T`.
 * Returns 1 if the option was consumed (2 if it also consumed `arg`), 0 if it is not one of ours. */
int gpuFaninHandleOption(struct gpu_fanin *f, const char *opt, const char *arg);
/* opens every file and creates the contexts; 0 or a negative MGPU_E_* code (loud: there is no CPU fallback) */
int gpuFaninOpen(struct gpu_fanin *f, gpu_stream_sink sink, void *user);
/* one reader thread per stream; returns when every stream reached EOF, failed, or gpuFaninCancel() was called.
 * 0 if every stream ended with MGPU_OK, else the first failing stream's code. */
int gpuFaninRun(struct gpu_fanin *f);
void gpuFaninCancel(struct gpu_fanin *f);
void gpuFaninClose(struct gpu_fanin *f);

#ifdef __cplusplus
}
#endif
#endif
