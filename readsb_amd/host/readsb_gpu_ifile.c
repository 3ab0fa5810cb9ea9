/* readsb_gpu_ifile — `readsb --device-type ifile --ifile X --iformat F --raw --mlat [--stats]`
 * with the demodulator on the GPU: prints one `@<12-hex 12 MHz timestamp><frame hex>;` line per
 * accepted message exactly as displayModesMessage does in --raw --mlat mode (mode_s.c:1834-1847),
 * and with --stats the demodulator counters of display_stats (stats.c:65-125).
 */
#include <fcntl.h>
#include <inttypes.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h>
#include <unistd.h>

#include "readsb_gpu_host.h"

static void print_raw_line(FILE *out, const struct gpu_modes_message *mm) {
    static const char hexl[] = "0123456789abcdef", hexu[] = "0123456789ABCDEF";
    char line[1 + 12 + 28 + 2], *p = line;
    *p++ = '@';
    for (int sh = 44; sh >= 0; sh -= 4) *p++ = hexu[((uint64_t) mm->timestamp >> sh) & 15];     /* %012 PRIX64 */
    for (int j = 0; j < mm->msgbits / 8; j++) { *p++ = hexl[mm->msg[j] >> 4]; *p++ = hexl[mm->msg[j] & 15]; }
    *p++ = ';'; *p++ = '\n';
    fwrite(line, 1, (size_t) (p - line), out);
}

static void print_raw(const struct gpu_modes_message *mm, void *user) { print_raw_line(user, mm); }

int main(int argc, char **argv) {
    struct mgpu_config cfg;
    mgpu_config_defaults(&cfg);
    const char *ifile = NULL;
    input_format_t fmt = INPUT_UC8;
    int stats = 0;
    unsigned chunk = 256;
    for (int i = 1; i < argc; i++) {
        if (!strcmp(argv[i], "--ifile") && i + 1 < argc) ifile = argv[++i];
        else if (!strcmp(argv[i], "--iformat") && i + 1 < argc) {
            const char *f = argv[++i];
            fmt = !strcasecmp(f, "UC8") ? INPUT_UC8 : !strcasecmp(f, "SC16") ? INPUT_SC16 : INPUT_SC16Q11;
        } else if (!strcmp(argv[i], "--fix")) cfg.nfix_crc = 1;
        else if (!strcmp(argv[i], "--no-fix")) cfg.nfix_crc = 0;
        else if (!strcmp(argv[i], "--aggressive")) cfg.nfix_crc = 2;
        else if (!strcmp(argv[i], "--no-fix-df")) cfg.fixDF = 0;
        else if (!strcmp(argv[i], "--modeac")) cfg.mode_ac = 1;        /* Modes.mode_ac, readsb.c:1479 */
        else if (!strcmp(argv[i], "--preamble-threshold") && i + 1 < argc) cfg.preamble_threshold = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--gpu-device") && i + 1 < argc) cfg.device = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--gpu-chunk-buffers") && i + 1 < argc) chunk = (unsigned) atoi(argv[++i]);
        else if (!strcmp(argv[i], "--startup-time-ms") && i + 1 < argc) cfg.startup_time_ms = atoll(argv[++i]);
        else if (!strcmp(argv[i], "--stats")) stats = 1;
        else if (!strcmp(argv[i], "--raw") || !strcmp(argv[i], "--mlat") || !strcmp(argv[i], "--quiet")) { }
        else if (!strcmp(argv[i], "--device-type") && i + 1 < argc) ++i;
        else { fprintf(stderr, "unknown option %s\n", argv[i]); return 2; }
    }
    if (!ifile) { fprintf(stderr, "SDR type 'ifile' requires an --ifile argument\n"); return 2; }   /* sdr_ifile.c:118 */
    int fd = !strcmp(ifile, "-") ? STDIN_FILENO : open(ifile, O_RDONLY);
    if (fd < 0) { perror(ifile); return 1; }
    cfg.format = (int) fmt;
    cfg.max_samples = (uint64_t) chunk * 131072;
    struct gpu_demod g;
    if (gpu_demod_open(&g, &cfg, print_raw, stdout) != MGPU_OK) return 1;
    int rc = gpu_ifile_run(&g, fd, fmt, chunk);
    if (rc != MGPU_OK) fprintf(stderr, "gpu_ifile_run: %s (%s)\n", mgpu_strerror(rc), mgpu_last_error(g.ctx));
    fflush(stdout);
    if (stats && rc == MGPU_OK) {
        const struct mgpu_counters *c = &g.counters;
        fprintf(stderr, "Local receiver:\n  %" PRIu64 " samples processed\n  %" PRIu64 " samples lost\n", c->samples_processed, c->samples_lost);
        fprintf(stderr, "  %" PRIu64 " Mode-S message preambles received\n", c->demod_preambles);
        fprintf(stderr, "    %" PRIu64 " with bad message format or invalid CRC\n", c->demod_rejected_bad);
        fprintf(stderr, "    %" PRIu64 " with unrecognized ICAO address\n", c->demod_rejected_unknown_icao);
        fprintf(stderr, "    %" PRIu64 " accepted with correct CRC\n", c->demod_accepted[0]);
        for (int i = 1; i <= 2; i++) fprintf(stderr, "    %" PRIu64 " accepted with %d-bit error repaired\n", c->demod_accepted[i], i);
        fprintf(stderr, "  %" PRIu64 " strong signals (> -3dBFS)\n", c->strong_signal_count);
    }
    gpu_demod_close(&g);
    if (fd != STDIN_FILENO) close(fd);
    return rc == MGPU_OK ? 0 : 1;
}
