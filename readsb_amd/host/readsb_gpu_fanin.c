/* readsb_gpu_fanin — N × (`readsb --device-type ifile --ifile X --raw --mlat`) in one process, every stream on its
 * own GPU demodulator context (sdr_gpu_fanin.c):
 *   readsb_gpu_fanin [--iformat F] --ifile A [--iformat F] --ifile B … --out-prefix P [--stats] [common options]
 * writes P.<stream> with one `@<12-hex 12 MHz timestamp><frame hex>;` line per accepted message of that stream
 * (displayModesMessage in --raw --mlat mode, mode_s.c:1834-1847), and with --stats one summary line per stream on stderr.
 */
#include <inttypes.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "readsb_gpu_host.h"

struct outputs { FILE **out; unsigned n; };

static void print_raw_line(FILE *out, const struct gpu_modes_message *mm) {
    static const char hexl[] = "0123456789abcdef", hexu[] = "0123456789ABCDEF";
    char line[1 + 12 + 28 + 2], *p = line;
    *p++ = '@';
    for (int sh = 44; sh >= 0; sh -= 4) *p++ = hexu[((uint64_t) mm->timestamp >> sh) & 15];     /* %012 PRIX64 */
    for (int j = 0; j < mm->msgbits / 8; j++) { *p++ = hexl[mm->msg[j] >> 4]; *p++ = hexl[mm->msg[j] & 15]; }
    *p++ = ';'; *p++ = '\n';
    fwrite(line, 1, (size_t) (p - line), out);
}

static void print_raw(unsigned stream, const struct gpu_modes_message *mm, void *user) {
    struct outputs *o = user;
    print_raw_line(o->out[stream], mm);         /* one file per stream, written only by that stream's thread */
}

int main(int argc, char **argv) {
    struct gpu_fanin f;
    gpuFaninInitConfig(&f);
    const char *prefix = NULL;
    int stats = 0;
    for (int i = 1; i < argc; i++) {
        const char *arg = i + 1 < argc ? argv[i + 1] : NULL;
        const int used = gpuFaninHandleOption(&f, argv[i], arg);
        if (used) { i += used - 1; continue; }
        if (!strcmp(argv[i], "--out-prefix") && arg) { prefix = arg; ++i; }
        else if (!strcmp(argv[i], "--stats")) stats = 1;
        else if (!strcmp(argv[i], "--raw") || !strcmp(argv[i], "--mlat") || !strcmp(argv[i], "--quiet")) { }
        else if (!strcmp(argv[i], "--device-type") && arg) ++i;
        else { fprintf(stderr, "unknown option %s\n", argv[i]); return 2; }
    }
    if (!prefix) { fprintf(stderr, "readsb_gpu_fanin: --out-prefix P is required (one output file per stream)\n"); return 2; }
    struct outputs o = { calloc(f.nstreams ? f.nstreams : 1, sizeof(FILE *)), f.nstreams };
    for (unsigned k = 0; k < f.nstreams; ++k) {
        char name[4096];
        snprintf(name, sizeof(name), "%s.%u", prefix, k);
        o.out[k] = fopen(name, "w");
        if (!o.out[k]) { perror(name); return 1; }
    }
    if (gpuFaninOpen(&f, print_raw, &o) != MGPU_OK) return 1;
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    const int rc = gpuFaninRun(&f);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    if (rc != MGPU_OK) fprintf(stderr, "gpuFaninRun: %s\n", mgpu_strerror(rc));
    uint64_t total = 0;
    for (unsigned k = 0; k < f.nstreams; ++k) {
        fclose(o.out[k]);
        total += f.streams[k].samples;
        if (stats) {
            const struct mgpu_counters *c = &f.streams[k].demod.counters;
            fprintf(stderr, "stream %u (%s, GPU %d): %" PRIu64 " samples processed, %" PRIu64 " Mode-S message preambles received, %" PRIu64
                    " accepted with correct CRC, %" PRIu64 " accepted with 1-bit error repaired\n", k, f.streams[k].path, f.streams[k].device,
                    c->samples_processed, c->demod_preambles, c->demod_accepted[0], c->demod_accepted[1]);
        }
    }
    if (stats) {
        const double s = (double) (t1.tv_sec - t0.tv_sec) + 1e-9 * (double) (t1.tv_nsec - t0.tv_nsec);
        fprintf(stderr, "fan-in: %u streams, %" PRIu64 " samples in %.3f s = %.1f Msamples/s (file reads and PCIe uploads included)\n",
                f.nstreams, total, s, (double) total / s / 1e6);
    }
    free(o.out);
    gpuFaninClose(&f);
    return rc == MGPU_OK ? 0 : 1;
}
