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
};

/* modesInit()'s hot-path part (readsb.c:285-310) + init_converter (sdr_ifile.c:156) */
int gpu_demod_open(struct gpu_demod *g, const struct mgpu_config *cfg, gpu_message_sink sink, void *user);
void gpu_demod_close(struct gpu_demod *g);

/* void demodulate2400(struct mag_buf *mag) (demod_2400.h:38): same call shape, decode thread only */
void demodulate2400_gpu(struct gpu_demod *g, struct mag_buf *mag);

/* Bulk path for file input: ifileRun's read loop (sdr_ifile.c:169-270) collapsed into large feeds:
 * reads `fd` to EOF in chunks of `chunk_buffers` 131072-sample buffers, converts and demodulates
 * on the GPU, delivers messages in stream order.  Returns 0 or a negative MGPU_E_* code. */
int gpu_ifile_run(struct gpu_demod *g, int fd, input_format_t format, unsigned chunk_buffers);

#ifdef __cplusplus
}
#endif
#endif
