/* readsb_gpu_shard — ONE capture time-chunked over the GPUs, in C (BASELINE.json configs[4]: "one long capture time-chunked ...
 * across the 8xMI355X node, with RCCL over xGMI used only to gather decoded-message lists/counts", host stays C).
 *
 * N processes, one per GPU, all reading the same sample file.  Every rank walks and builds its OWN range of whole buffers; what ties
 * the ranges together — the ICAO filter's state and the data-dependent clock of its 60 s expiry (readsb.c:1227-1231,
 * demod_2400.c:412-414, icao_filter.c:65-130) — is settled by the protocol of include/modes_gpu.h ("config 5 with the ordered walk
 * itself sharded", stream form; DESIGN.md §5 says why it is exact).  The same call sequence as readsb_amd/shard.py
 * (demodulate_sharded_stream), without Python:
 *   1. pre-pass   the buffers of my range an expiry can follow (mgpu_expiry_windows), gathered HBM -> HBM into one short stream,
 *                 their end clocks estimated from the records (mgpu_shard_clock_estimate); all-gather; the schedule (mgpu_flip_schedule);
 *   2. pass       two filter generations of warm-up + my range through the ordinary pipeline, schedule imposed (mgpu_shard_stream_*);
 *   3. round      all-gather of true end clocks + the filter state at my range's two ends; mgpu_shard_round: done, or the new schedule
 *                 and whose end state I start my next pass from;
 *   4. result     message counts, counters, noise terms, sum blocks and the 64-byte records of every rank -> rank 0, which adds the
 *                 integer counters, re-adds the two sequential double sums in stream order (mgpu_seqsum, mgpu_seqsum_apply), encodes
 *                 one beast stream on its GPU (--out) and prints the counters as one JSON line on stdout.
 * Transport: RCCL (--id-file: rank 0 writes the ncclUniqueId there, as readsb_gpu_gather does), or files in a directory
 * (--file-transport DIR: every blob a file, for dry runs with several ranks on ONE GPU — the pool's boxes have one).
 *
 *   readsb_gpu_shard --rank R --world N (--id-file PATH | --file-transport DIR) --ifile capture.iq [--iformat UC8|SC16|SC16Q11]
 *                    [--fix|--no-fix|--aggressive] [--no-fix-df] [--preamble-threshold T] [--startup-time-ms T] [--gpu-device D] [--out beast.bin]
 */
#define __HIP_PLATFORM_AMD__ 1
#include <fcntl.h>
#include <inttypes.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include "../../include/modes_gpu.h"

#define CHK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
#define CHK_NCCL(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) { fprintf(stderr, "%s: %s\n", #x, ncclGetErrorString(r_)); return 1; } } while (0)
#define CHK_MGPU(x, ctx) do { int r_ = (x); if (r_ != MGPU_OK) { fprintf(stderr, "%s: %s (%s)\n", #x, mgpu_strerror(r_), (ctx) ? mgpu_last_error(ctx) : ""); return 1; } } while (0)
#define CHK(x) do { if ((x) != 0) { fprintf(stderr, "%s failed\n", #x); return 1; } } while (0)

enum { BUF = 131072, TRAILING = 326, SUM_BLOCK = 1024 };

/* ---- transport: variable-size all-gather of host blobs, and the records' gather to rank 0 ---- */
struct transport {
    int rank, world, use_files, seq;
    char dir[3072];
    char run[64];                 /* --run-id: part of every file name, so that a directory (or id-file) left over from another run is never read */
    ncclComm_t comm;
    hipStream_t s;
};

static void file_name(const struct transport *t, int seq, int rank, char *out, size_t cap) {
    snprintf(out, cap, "%s/x%s.%d.%d", t->dir, t->run, seq, rank);
}

static int file_put(const struct transport *t, int seq, const void *p, uint64_t bytes) {
    char tmp[4300], fin[4200];
    file_name(t, seq, t->rank, fin, sizeof(fin));
    snprintf(tmp, sizeof(tmp), "%s.tmp", fin);
    FILE *f = fopen(tmp, "wb");
    if (!f) { perror(tmp); return -1; }
    const int bad = bytes && fwrite(p, 1, bytes, f) != bytes;
    if (fclose(f) != 0 || bad) { perror(tmp); unlink(tmp); return -1; }
    return rename(tmp, fin);
}

/* At exit — behind a last, EMPTY all-gather (main): finishing that one proves every peer has been through the exchange before it,
 * i.e. has read everything this rank wrote up to the gather's payload (the largest file of the run: messages + blocks + terms).  A
 * rank then removes all its files but the barrier's own empty one, which a slower peer may still be polling for. */
static void file_cleanup(const struct transport *t) {
    if (!t->use_files) return;
    for (int seq = 0; seq + 1 < t->seq; ++seq) {
        char fin[4200];
        file_name(t, seq, t->rank, fin, sizeof(fin));
        unlink(fin);
    }
}

static int file_get(const struct transport *t, int seq, int from, void **p, uint64_t *bytes) {
    char fin[4200];
    file_name(t, seq, from, fin, sizeof(fin));
    for (int tries = 0; tries < 600000; ++tries) {                     /* up to ten minutes */
        struct stat st;
        if (stat(fin, &st) == 0) {
            FILE *f = fopen(fin, "rb");
            if (!f) return -1;
            *bytes = (uint64_t) st.st_size;
            *p = malloc(*bytes + 8);
            const size_t got = *bytes ? fread(*p, 1, *bytes, f) : 0;
            fclose(f);
            return got == *bytes ? 0 : -1;
        }
        struct timespec ts = {0, 1000000};
        nanosleep(&ts, NULL);
    }
    fprintf(stderr, "rank %d: nothing from rank %d (%s)\n", t->rank, from, fin);
    return -1;
}

/* all[r] (malloc'd) = rank r's blob, sizes[r] its length */
static int t_allgather(struct transport *t, const void *mine, uint64_t bytes, void **all, uint64_t *sizes) {
    const int seq = t->seq++;
    if (t->use_files) {
        if (file_put(t, seq, mine, bytes) != 0) return 1;
        for (int r = 0; r < t->world; ++r)
            if (file_get(t, seq, r, &all[r], &sizes[r]) != 0) return 1;
        return 0;
    }
    unsigned long long *d_sz = NULL, my = bytes;
    CHK_HIP(hipMalloc((void **) &d_sz, (size_t) (t->world + 1) * sizeof(*d_sz)));
    CHK_HIP(hipMemcpyAsync(d_sz + t->world, &my, sizeof(my), hipMemcpyHostToDevice, t->s));
    CHK_NCCL(ncclAllGather(d_sz + t->world, d_sz, 1, ncclUint64, t->comm, t->s));
    unsigned long long *sz = malloc((size_t) t->world * sizeof(*sz));
    CHK_HIP(hipMemcpyAsync(sz, d_sz, (size_t) t->world * sizeof(*sz), hipMemcpyDeviceToHost, t->s));
    CHK_HIP(hipStreamSynchronize(t->s));
    uint64_t mx = 8;
    for (int r = 0; r < t->world; ++r) { sizes[r] = sz[r]; if (sz[r] > mx) mx = sz[r]; }
    mx = (mx + 7) & ~7ull;
    uint8_t *d_buf = NULL, *h = malloc(mx * (size_t) t->world);
    CHK_HIP(hipMalloc((void **) &d_buf, mx * (size_t) (t->world + 1)));
    if (bytes) CHK_HIP(hipMemcpyAsync(d_buf + mx * (size_t) t->world, mine, bytes, hipMemcpyHostToDevice, t->s));
    CHK_NCCL(ncclAllGather(d_buf + mx * (size_t) t->world, d_buf, mx, ncclUint8, t->comm, t->s));
    CHK_HIP(hipMemcpyAsync(h, d_buf, mx * (size_t) t->world, hipMemcpyDeviceToHost, t->s));
    CHK_HIP(hipStreamSynchronize(t->s));
    for (int r = 0; r < t->world; ++r) { all[r] = malloc(sizes[r] + 8); memcpy(all[r], h + mx * (size_t) r, sizes[r]); }
    free(h); free(sz);
    (void) hipFree(d_buf); (void) hipFree(d_sz);
    return 0;
}

/* rank 0: all[r] = rank r's blob (sizes known to everybody beforehand); the others: nothing */
static int t_gather_root(struct transport *t, const void *mine, uint64_t bytes, const uint64_t *sizes, void **all) {
    const int seq = t->seq++;
    if (t->use_files) {
        if (file_put(t, seq, mine, bytes) != 0) return 1;
        if (t->rank == 0)
            for (int r = 0; r < t->world; ++r) { uint64_t got = 0; if (file_get(t, seq, r, &all[r], &got) != 0 || got != sizes[r]) return 1; }
        return 0;
    }
    uint8_t *d_mine = NULL, *d_all = NULL;
    uint64_t total = 0;
    for (int r = 0; r < t->world; ++r) total += sizes[r];
    CHK_HIP(hipMalloc((void **) &d_mine, bytes + 8));
    if (bytes) CHK_HIP(hipMemcpyAsync(d_mine, mine, bytes, hipMemcpyHostToDevice, t->s));
    if (t->rank == 0) CHK_HIP(hipMalloc((void **) &d_all, total + 8));
    CHK_NCCL(ncclGroupStart());
    if (t->rank == 0) {
        uint64_t off = 0;
        for (int r = 0; r < t->world; ++r) { if (sizes[r]) CHK_NCCL(ncclRecv(d_all + off, sizes[r], ncclUint8, r, t->comm, t->s)); off += sizes[r]; }
    }
    if (bytes) CHK_NCCL(ncclSend(d_mine, bytes, ncclUint8, 0, t->comm, t->s));
    CHK_NCCL(ncclGroupEnd());
    CHK_HIP(hipStreamSynchronize(t->s));
    if (t->rank == 0) {
        uint64_t off = 0;
        for (int r = 0; r < t->world; ++r) { all[r] = malloc(sizes[r] + 8); CHK_HIP(hipMemcpy(all[r], d_all + off, sizes[r], hipMemcpyDeviceToHost)); off += sizes[r]; }
        (void) hipFree(d_all);
    }
    (void) hipFree(d_mine);
    return 0;
}

static int exchange_id(const char *path0, const char *run, int rank, ncclUniqueId *id) {
    char path[4200];
    snprintf(path, sizeof(path), "%s%s%s", path0, run[0] ? "." : "", run);   /* (the run id: an id-file of an earlier run is another file) */
    if (rank == 0) {
        if (ncclGetUniqueId(id) != ncclSuccess) return -1;
        char tmp[4300];
        snprintf(tmp, sizeof(tmp), "%s.tmp", path);
        unlink(path);                                                          /* (same run id twice: the old id goes before the new one is written) */
        FILE *f = fopen(tmp, "wb");
        if (!f) { perror(tmp); return -1; }
        const int bad = fwrite(id, sizeof(*id), 1, f) != 1;
        if (fclose(f) != 0 || bad) { perror(tmp); unlink(tmp); return -1; }
        return rename(tmp, path);
    }
    for (int tries = 0; tries < 60000; ++tries) {
        FILE *f = fopen(path, "rb");
        if (f) { const size_t got = fread(id, sizeof(*id), 1, f); fclose(f); if (got == 1) return 0; }
        struct timespec ts = {0, 1000000};
        nanosleep(&ts, NULL);
    }
    return -1;
}

static int64_t sys_ms(uint64_t b, int64_t startup) { return (int64_t) ((b * (uint64_t) BUF * 5) / 12000) + startup; }

/* feed samples [a, b) of the capture (device-resident from sample `lo` on) in calls of at most `cap` samples */
static int feed(mgpu_ctx *ctx, const uint8_t *d_iq, uint64_t lo, size_t bps, uint64_t a, uint64_t b, uint64_t cap) {
    for (uint64_t off = a; off < b; off += cap) {
        const uint64_t k = b - off < cap ? b - off : cap;
        CHK_MGPU(mgpu_feed_iq_device(ctx, d_iq + (off - lo) * bps, k), ctx);
    }
    return 0;
}

/* The exchange's stream with a hardware queue of its own (hipExtStreamCreateWithCUMask, every CU enabled): an ordinary stream shares the
 * runtime's small pool of queues with the demodulator's main stream, and a collective that waits for its peers at the head of a shared queue
 * holds the chunk's kernels behind it (DESIGN.md §4 "The side streams' queues").  Falls back to an ordinary stream. */
static hipError_t own_queue_stream(hipStream_t *s) {
    int dev = 0;
    hipDeviceProp_t prop;
    uint32_t mask[32];
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0 && prop.multiProcessorCount <= 1024) {
        memset(mask, 0, sizeof(mask));
        for (int cu = 0; cu < prop.multiProcessorCount; ++cu) mask[cu >> 5] |= 1u << (cu & 31);
        if (hipExtStreamCreateWithCUMask(s, (uint32_t) ((prop.multiProcessorCount + 31) / 32), mask) == hipSuccess) return hipSuccess;
        (void) hipGetLastError();
    }
    return hipStreamCreate(s);
}

int main(int argc, char **argv) {
    struct mgpu_config cfg;
    mgpu_config_defaults(&cfg);
    const char *ifile = NULL, *idfile = NULL, *outpath = NULL, *tdir = NULL, *runid = "";
    int rank = 0, world = 1, device = -1;
    for (int i = 1; i < argc; i++) {
        if (!strcmp(argv[i], "--ifile") && i + 1 < argc) ifile = argv[++i];
        else if (!strcmp(argv[i], "--iformat") && i + 1 < argc) {
            const char *f = argv[++i];
            cfg.format = !strcasecmp(f, "UC8") ? MGPU_FMT_UC8 : !strcasecmp(f, "SC16") ? MGPU_FMT_SC16 : MGPU_FMT_SC16Q11;
        } else if (!strcmp(argv[i], "--fix")) cfg.nfix_crc = 1;
        else if (!strcmp(argv[i], "--no-fix")) cfg.nfix_crc = 0;
        else if (!strcmp(argv[i], "--aggressive")) cfg.nfix_crc = 2;
        else if (!strcmp(argv[i], "--no-fix-df")) cfg.fixDF = 0;
        else if (!strcmp(argv[i], "--preamble-threshold") && i + 1 < argc) cfg.preamble_threshold = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--startup-time-ms") && i + 1 < argc) cfg.startup_time_ms = atoll(argv[++i]);
        else if (!strcmp(argv[i], "--gpu-device") && i + 1 < argc) device = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--rank") && i + 1 < argc) rank = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--world") && i + 1 < argc) world = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--id-file") && i + 1 < argc) idfile = argv[++i];
        else if (!strcmp(argv[i], "--file-transport") && i + 1 < argc) tdir = argv[++i];
        else if (!strcmp(argv[i], "--run-id") && i + 1 < argc) runid = argv[++i];
        else if (!strcmp(argv[i], "--out") && i + 1 < argc) outpath = argv[++i];
        else { fprintf(stderr, "unknown option %s\n", argv[i]); return 2; }
    }
    if (!ifile || (!idfile && !tdir) || world < 1 || rank < 0 || rank >= world) {
        fprintf(stderr, "usage: %s --rank R --world N (--id-file PATH | --file-transport DIR) [--run-id ID (the same on every rank, new for every run)] --ifile FILE [--iformat F] [--fix|--no-fix|--aggressive] [--out beast.bin]\n", argv[0]);
        return 2;
    }
    cfg.device = device >= 0 ? device : (tdir ? 0 : rank);
    CHK_HIP(hipSetDevice(cfg.device));
    struct transport T = {rank, world, tdir != NULL, 0, {0}, {0}, NULL, NULL};
    snprintf(T.run, sizeof(T.run), "%s", runid);
    if (tdir) snprintf(T.dir, sizeof(T.dir), "%s", tdir);
    else {
        ncclUniqueId id;
        if (exchange_id(idfile, T.run, rank, &id) != 0) { fprintf(stderr, "rank %d: no ncclUniqueId\n", rank); return 1; }
        CHK_NCCL(ncclCommInitRank(&T.comm, world, id, rank));
    }
    CHK_HIP(own_queue_stream(&T.s));

    /* ---- the capture, my range of whole buffers, the warm-up before it ---- */
    const int fd = open(ifile, O_RDONLY);
    if (fd < 0) { perror(ifile); return 1; }
    struct stat st;
    if (fstat(fd, &st) != 0) { perror(ifile); return 1; }
    const size_t bps = cfg.format == MGPU_FMT_UC8 ? 2 : 4;
    const uint64_t n = (uint64_t) st.st_size / bps, nbuf_total = (n + BUF - 1) / BUF;
    const uint8_t *iq = n ? mmap(NULL, (size_t) st.st_size, PROT_READ, MAP_PRIVATE, fd, 0) : NULL;
    if (n && iq == MAP_FAILED) { perror("mmap"); return 1; }
    const uint64_t b0 = nbuf_total * (uint64_t) rank / (uint64_t) world, b1 = nbuf_total * (uint64_t) (rank + 1) / (uint64_t) world;
    const uint64_t first = b0 * BUF < n ? b0 * BUF : n, last = b1 * BUF < n ? b1 * BUF : n;
    const uint64_t nbuf_own = (last - first + BUF - 1) / BUF;
    /* two filter generations in whole buffers; a generation lasts up to 60 s + two buffers */
    const uint64_t warmup = ((2ull * 60 * 2400000 + BUF - 1) / BUF) * BUF + 5ull * BUF;
    const uint64_t ws = first > warmup ? first - warmup : 0;
    const uint64_t lo = ws > TRAILING ? ws - TRAILING : 0;              /* first sample this rank keeps in HBM */
    uint8_t *d_iq = NULL;
    CHK_HIP(hipMalloc((void **) &d_iq, (last - lo) * bps + 64));
    if (last > lo) CHK_HIP(hipMemcpy(d_iq, iq + lo * bps, (last - lo) * bps, hipMemcpyHostToDevice));
    const uint64_t cap = 4096ull * BUF;                                 /* samples per feed call */
    cfg.max_samples = cap;
    mgpu_ctx *ctx = NULL;
    CHK_MGPU(mgpu_create(&cfg, &ctx), ctx);
    const int64_t startup = cfg.startup_time_ms;
    const int fc = (int) cfg.filter_clock;

    /* ---- 1. pre-pass: my buffers inside the expiry windows, as one short stream; their end clocks, estimated ---- */
    uint8_t *mask = calloc(nbuf_total + 1, 1);
    (void) mgpu_expiry_windows(nbuf_total, BUF, startup, fc, mask);
    uint64_t nwin = 0;
    int64_t *widx = malloc((nbuf_own + 1) * sizeof(*widx)), *wclk = malloc((nbuf_own + 1) * sizeof(*wclk));
    for (uint64_t b = b0; b < b0 + nbuf_own; ++b)
        if (mask[b] && !(b == nbuf_total - 1 && n % BUF)) widx[nwin++] = (int64_t) b;   /* (a short last buffer does not fit the gathered grid: its clock is guessed) */
    if (nwin) {
        uint8_t *d_gather = NULL;
        CHK_HIP(hipMalloc((void **) &d_gather, nwin * BUF * bps + 64));
        for (uint64_t k = 0; k < nwin;) {                               /* runs of consecutive buffers: one copy each */
            uint64_t e = k + 1;
            while (e < nwin && widx[e] == widx[e - 1] + 1) ++e;
            CHK_HIP(hipMemcpyAsync(d_gather + k * BUF * bps, d_iq + ((uint64_t) widx[k] * BUF - lo) * bps, (e - k) * BUF * bps, hipMemcpyDeviceToDevice, T.s));
            k = e;
        }
        CHK_HIP(hipStreamSynchronize(T.s));
        CHK_MGPU(mgpu_reset(ctx), ctx);
        CHK_MGPU(mgpu_shard_begin(ctx, 0, NULL, 3), ctx);          /* clock estimates only */
        CHK(feed(ctx, d_gather, 0, bps, 0, nwin * BUF, cap));
        uint64_t got = 0;
        CHK_MGPU(mgpu_shard_clock_estimate(ctx, NULL, 0, 0, wclk, nwin + 1, &got), ctx);
        if (got != nwin) { fprintf(stderr, "pre-pass: %" PRIu64 " clocks for %" PRIu64 " buffers\n", got, nwin); return 1; }
        /* the gathered stream's buffer k stands for buffer widx[k]: what counts is how far into its own 55 ms the clock ends */
        for (uint64_t k = 0; k < nwin; ++k) wclk[k] = sys_ms((uint64_t) widx[k], startup) + (wclk[k] - sys_ms(k, startup));
        (void) hipFree(d_gather);
    }
    void **all = calloc((size_t) world, sizeof(*all));
    uint64_t *sizes = calloc((size_t) world, sizeof(*sizes));
    int64_t *clocks_all = malloc((nbuf_total + 2) * sizeof(*clocks_all));
    for (uint64_t b = 0; b < nbuf_total; ++b) clocks_all[b] = sys_ms(b, startup);      /* outside the windows no expiry can follow: any clock within the buffer does */
    {
        int64_t *blob = malloc((2 * nwin + 1) * sizeof(*blob));
        memcpy(blob, widx, nwin * sizeof(*blob));
        memcpy(blob + nwin, wclk, nwin * sizeof(*blob));
        CHK(t_allgather(&T, blob, 2 * nwin * sizeof(*blob), all, sizes));
        free(blob);
        for (int r = 0; r < world; ++r) {
            const uint64_t k = sizes[r] / 16;
            const int64_t *p = all[r];
            for (uint64_t i = 0; i < k; ++i) if ((uint64_t) p[i] < nbuf_total) clocks_all[p[i]] = p[k + i];
            free(all[r]);
        }
    }
    uint64_t nclk = nbuf_total;
    if (n % BUF == 0) clocks_all[nclk++] = (int64_t) ((n * 5) / 12000) + startup;      /* the EOF buffer (sdr_ifile.c:223-237) */
    const uint64_t sched_cap = nclk / 1000 + 64;
    uint64_t *fl = malloc(sched_cap * sizeof(*fl));
    int64_t *sched = malloc(sched_cap * sizeof(*sched)), *next = malloc(sched_cap * sizeof(*next));
    uint64_t nsched = mgpu_flip_schedule(clocks_all, nclk, startup, fc, fl, sched_cap);
    if (nsched > sched_cap) { fprintf(stderr, "schedule longer than expected\n"); return 1; }
    for (uint64_t i = 0; i < nsched; ++i) sched[i] = (int64_t) (fl[i] * BUF) * 5;

    /* ---- 2. + 3. the pass and the rounds ---- */
    uint64_t msg_cap = (last - first) / 64 + 65536, nmsg = 0, nterms = 0, nclocks = 0;
    struct mgpu_msg *msgs = malloc(msg_cap * sizeof(*msgs));
    struct mgpu_counters counters;
    memset(&counters, 0, sizeof(counters));
    int64_t *clocks = malloc((nbuf_own + 2) * sizeof(*clocks));
    double *terms = malloc((nbuf_own + 2) * sizeof(*terms));
    void *sf = NULL, *se = NULL, *import = NULL;
    uint64_t sfb = 0, seb = 0, importb = 0;
    int rounds = 0, passes = 0, have_pass = 0;
    int64_t *used = malloc(sched_cap * sizeof(*used));
    uint64_t nused = 0;
    int used_import = 0;
    for (;;) {
        if (++rounds > world + 72) { fprintf(stderr, "the schedule / seam rounds did not settle\n"); return 1; }
        /* my pass, unless nothing it depends on has changed: the schedule inside [ws, last), the expiries before my range, the state I start from */
        int need = !have_pass || used_import != (import != NULL);
        if (!need) {
            uint64_t i = 0, j = 0, before_a = 0, before_b = 0;
            for (uint64_t q = 0; q < nsched; ++q) before_a += sched[q] < (int64_t) first * 5;
            for (uint64_t q = 0; q < nused; ++q) before_b += used[q] < (int64_t) first * 5;
            need = before_a != before_b;
            while (!need) {                                                  /* the entries inside my window, pairwise */
                while (i < nsched && (sched[i] < (int64_t) ws * 5 || sched[i] >= (int64_t) last * 5)) ++i;
                while (j < nused && (used[j] < (int64_t) ws * 5 || used[j] >= (int64_t) last * 5)) ++j;
                if (i >= nsched || j >= nused) { need = (i < nsched) != (j < nused); break; }
                if (sched[i] != used[j]) need = 1;
                ++i; ++j;
            }
        }
        if (need && nbuf_own) {
            const uint64_t start = import ? first : ws;
            struct mgpu_shard_stream_args a = {start, start ? iq + (start - TRAILING) * bps : NULL, first, sched, nsched, import, importb};
            CHK_MGPU(mgpu_shard_stream_begin(ctx, &a), ctx);
            CHK(feed(ctx, d_iq, lo, bps, start, first, cap));               /* the warm-up (nothing with an imported state) */
            CHK_MGPU(mgpu_shard_stream_mark(ctx), ctx);
            nmsg = 0;
            for (uint64_t off = first; off < last; off += cap) {            /* the range; every call's messages land behind the previous call's */
                const uint64_t k = last - off < cap ? last - off : cap;
                CHK_MGPU(mgpu_set_message_buffer(ctx, msgs + nmsg, msg_cap - nmsg), ctx);
                CHK_MGPU(mgpu_feed_iq_device(ctx, d_iq + (off - lo) * bps, k), ctx);
                uint64_t got = 0;
                CHK_MGPU(mgpu_collect(ctx, msgs + nmsg, msg_cap - nmsg, &got, NULL), ctx);
                nmsg += got;
            }
            CHK_MGPU(mgpu_shard_stream_end(ctx, clocks, nbuf_own + 1, &nclocks), ctx);
            uint64_t late = 0;                                              /* (the feeds above are synchronous: nothing is left — but if something were, it counts) */
            CHK_MGPU(mgpu_collect(ctx, msgs + nmsg, msg_cap - nmsg, &late, &counters), ctx);
            nmsg += late;
            const void *p0, *p1;
            const double *tp;
            CHK_MGPU(mgpu_shard_state(ctx, 0, &p0, &sfb), ctx);
            CHK_MGPU(mgpu_shard_state(ctx, 1, &p1, &seb), ctx);
            free(sf); free(se);
            sf = malloc(sfb + 8); se = malloc(seb + 8);
            memcpy(sf, p0, sfb); memcpy(se, p1, seb);
            CHK_MGPU(mgpu_shard_noise_terms(ctx, &tp, &nterms), ctx);
            memcpy(terms, tp, nterms * sizeof(*terms));
            memcpy(used, sched, nsched * sizeof(*used));
            nused = nsched;
            used_import = import != NULL;
            have_pass = 1;
            ++passes;
        } else if (!nbuf_own) have_pass = 1;
        /* the round: [nclocks | sfb | seb | clocks | state_first | state_end] from everybody */
        const uint64_t bl = 24 + nclocks * 8 + sfb + seb;
        uint8_t *blob = malloc(bl + 8);
        const uint64_t head[3] = {nclocks, sfb, seb};
        memcpy(blob, head, 24);
        memcpy(blob + 24, clocks, nclocks * 8);
        if (sfb) memcpy(blob + 24 + nclocks * 8, sf, sfb);
        if (seb) memcpy(blob + 24 + nclocks * 8 + sfb, se, seb);
        CHK(t_allgather(&T, blob, bl, all, sizes));
        free(blob);
        const int64_t **cl = malloc((size_t) world * sizeof(*cl));
        const void **s0 = malloc((size_t) world * sizeof(*s0)), **s1 = malloc((size_t) world * sizeof(*s1));
        uint64_t *ncl = malloc((size_t) world * 8), *n0 = malloc((size_t) world * 8), *n1 = malloc((size_t) world * 8);
        for (int r = 0; r < world; ++r) {
            const uint8_t *p = all[r];
            uint64_t h[3];
            memcpy(h, p, 24);
            ncl[r] = h[0]; n0[r] = h[1]; n1[r] = h[2];
            cl[r] = (const int64_t *) (p + 24);
            s0[r] = p + 24 + h[0] * 8; s1[r] = p + 24 + h[0] * 8 + h[1];
        }
        uint64_t nnext = 0;
        int32_t done = 0, *imp = malloc((size_t) world * sizeof(*imp));
        CHK_MGPU(mgpu_shard_round(sched, nsched, (uint32_t) world, cl, ncl, s0, n0, s1, n1, n, BUF, startup, fc, next, sched_cap, &nnext, imp, &done), ctx);
        if (!done) {
            memcpy(sched, next, nnext * sizeof(*sched));
            nsched = nnext;
            if (imp[rank] >= 0) {                                           /* my seam failed: my next pass starts from my predecessor's end state */
                free(import);
                importb = n1[imp[rank]];
                import = malloc(importb + 8);
                memcpy(import, s1[imp[rank]], importb);
                have_pass = 0;
            }
        }
        for (int r = 0; r < world; ++r) free(all[r]);
        free(cl); free(s0); free(s1); free(ncl); free(n0); free(n1); free(imp);
        if (done) break;
    }

    /* ---- 4. the result on rank 0 ---- */
    struct meta { uint64_t nmsg, nterms; struct mgpu_counters c; } me;
    memset(&me, 0, sizeof(me));
    me.nmsg = nmsg; me.nterms = nterms; me.c = counters;
    CHK(t_allgather(&T, &me, sizeof(me), all, sizes));
    struct meta *metas = malloc((size_t) world * sizeof(*metas));
    double approx = 0;
    for (int r = 0; r < world; ++r) { memcpy(&metas[r], all[r], sizeof(me)); free(all[r]); if (r < rank) approx += metas[r].c.signal_power_sum; }
    const uint64_t nblk = (nmsg + SUM_BLOCK - 1) / SUM_BLOCK;
    struct mgpu_sum_block *blocks = malloc((nblk + 1) * sizeof(*blocks));
    {   /* every rank at once: its part of the sequential signal-power sum — from the 8-byte numerators the builder logged during the pass */
        const uint64_t *sig_terms = NULL;
        uint64_t nsig = 0;
        CHK_MGPU(mgpu_shard_signal_terms(ctx, &sig_terms, &nsig), ctx);
        if (nsig == nmsg && nmsg) CHK_MGPU(mgpu_seqsum_blocks_terms(approx, sig_terms, nmsg, SUM_BLOCK, blocks), ctx);
        else CHK_MGPU(mgpu_seqsum_blocks(approx, msgs, nmsg, SUM_BLOCK, blocks), ctx);
    }
    const uint64_t mine_bytes = nmsg * sizeof(*msgs) + nblk * sizeof(*blocks) + nterms * sizeof(*terms);
    uint8_t *mine = malloc(mine_bytes + 8);
    memcpy(mine, msgs, nmsg * sizeof(*msgs));
    memcpy(mine + nmsg * sizeof(*msgs), blocks, nblk * sizeof(*blocks));
    memcpy(mine + nmsg * sizeof(*msgs) + nblk * sizeof(*blocks), terms, nterms * sizeof(*terms));
    for (int r = 0; r < world; ++r) sizes[r] = metas[r].nmsg * sizeof(*msgs) + ((metas[r].nmsg + SUM_BLOCK - 1) / SUM_BLOCK) * sizeof(*blocks) + metas[r].nterms * sizeof(*terms);
    CHK(t_gather_root(&T, mine, mine_bytes, sizes, all));
    int rc = 0;
    if (rank == 0) {
        struct mgpu_counters k;
        memset(&k, 0, sizeof(k));
        uint64_t total = 0;
        for (int r = 0; r < world; ++r) total += metas[r].nmsg;
        struct mgpu_msg *allm = malloc((total + 1) * sizeof(*allm));
        uint64_t off = 0, fallbacks = 0, nblocks = 0;
        double sig = 0, noise = 0;
        for (int r = 0; r < world; ++r) {
            const struct mgpu_counters *c = &metas[r].c;
            const uint64_t nm = metas[r].nmsg, nb = (nm + SUM_BLOCK - 1) / SUM_BLOCK;
            const uint8_t *p = all[r];
            if (nm || metas[r].nterms) {
                k.demod_preambles += c->demod_preambles; k.demod_rejected_bad += c->demod_rejected_bad; k.demod_rejected_unknown_icao += c->demod_rejected_unknown_icao;
                for (int i = 0; i < 3; ++i) k.demod_accepted[i] += c->demod_accepted[i];
                for (int i = 0; i < 5; ++i) { k.demod_preamblePhase[i] += c->demod_preamblePhase[i]; k.demod_bestPhase[i] += c->demod_bestPhase[i]; }
                k.strong_signal_count += c->strong_signal_count; k.signal_power_count += c->signal_power_count; k.noise_power_count += c->noise_power_count;
                k.samples_processed += c->samples_processed; k.samples_lost += c->samples_lost; k.nbuffers += c->nbuffers;
                if (c->peak_signal_power > k.peak_signal_power) k.peak_signal_power = c->peak_signal_power;
            }
            /* the two sequential double sums of the reference (demod_2400.c:445-447, 474-479), re-added in stream order */
            uint64_t fb = 0;
            sig = mgpu_seqsum_apply(sig, (const struct mgpu_msg *) p, nm, SUM_BLOCK, (const struct mgpu_sum_block *) (p + nm * sizeof(*msgs)), &fb);
            noise = mgpu_seqsum(noise, (const double *) (p + nm * sizeof(*msgs) + nb * sizeof(*blocks)), metas[r].nterms);
            fallbacks += fb; nblocks += nb;
            memcpy(allm + off, p, nm * sizeof(*msgs));
            off += nm;
            free(all[r]);
        }
        if (n % BUF == 0) { noise += (double) NAN; k.samples_lost += BUF; k.nbuffers += 1; }      /* the EOF buffer: 0 / 0 in the converter (convert.c:101-107) */
        k.nflips = nsched + (fc == MGPU_FILTER_CLOCK_BEFORE_FIRST ? 1 : 0);
        k.signal_power_sum = sig; k.noise_power_sum = noise;
        if (outpath) {
            uint64_t bytes = 0;
            const uint64_t out_cap = total * 48 + 64;
            uint8_t *out = malloc(out_cap);
            CHK_MGPU(mgpu_beast_encode(ctx, allm, total, out, out_cap, &bytes), ctx);
            FILE *f = fopen(outpath, "wb");
            if (!f) { perror(outpath); return 1; }
            if (bytes && fwrite(out, 1, bytes, f) != bytes) rc = 1;
            fclose(f);
            free(out);
        }
        uint64_t bits[3];                                                    /* the doubles as their bit patterns: compared with == by the test */
        memcpy(&bits[0], &k.signal_power_sum, 8); memcpy(&bits[1], &k.noise_power_sum, 8); memcpy(&bits[2], &k.peak_signal_power, 8);
        printf("{\"ranks\": %d, \"messages\": %" PRIu64 ", \"rounds\": %d, \"expiries\": %" PRIu64 ", \"sum_blocks\": %" PRIu64 ", \"sum_blocks_readded\": %" PRIu64
               ", \"demod_preambles\": %" PRIu64 ", \"demod_rejected_bad\": %" PRIu64 ", \"demod_rejected_unknown_icao\": %" PRIu64
               ", \"demod_accepted\": [%" PRIu64 ", %" PRIu64 ", %" PRIu64 "], \"demod_preamblePhase\": [%" PRIu64 ", %" PRIu64 ", %" PRIu64 ", %" PRIu64 ", %" PRIu64 "]"
               ", \"demod_bestPhase\": [%" PRIu64 ", %" PRIu64 ", %" PRIu64 ", %" PRIu64 ", %" PRIu64 "], \"strong_signal_count\": %" PRIu64
               ", \"signal_power_count\": %" PRIu64 ", \"noise_power_count\": %" PRIu64 ", \"samples_processed\": %" PRIu64 ", \"samples_lost\": %" PRIu64
               ", \"nbuffers\": %" PRIu64 ", \"nflips\": %" PRIu64 ", \"signal_power_sum_bits\": %" PRIu64 ", \"noise_power_sum_bits\": %" PRIu64
               ", \"peak_signal_power_bits\": %" PRIu64 "}\n",
               world, total, rounds, nsched, nblocks, fallbacks, k.demod_preambles, k.demod_rejected_bad, k.demod_rejected_unknown_icao,
               k.demod_accepted[0], k.demod_accepted[1], k.demod_accepted[2], k.demod_preamblePhase[0], k.demod_preamblePhase[1], k.demod_preamblePhase[2],
               k.demod_preamblePhase[3], k.demod_preamblePhase[4], k.demod_bestPhase[0], k.demod_bestPhase[1], k.demod_bestPhase[2], k.demod_bestPhase[3],
               k.demod_bestPhase[4], k.strong_signal_count, k.signal_power_count, k.noise_power_count, k.samples_processed, k.samples_lost, k.nbuffers, k.nflips,
               bits[0], bits[1], bits[2]);
        free(allm);
    }
    fprintf(stderr, "readsb_gpu_shard: rank %d of %d: buffers %" PRIu64 "..%" PRIu64 ", %" PRIu64 " pre-pass buffers, %d round(s), %d pass(es)%s, %" PRIu64 " messages\n",
            rank, world, b0, b0 + nbuf_own, nwin, rounds, passes, import ? " (imported state)" : "", nmsg);
    mgpu_destroy(ctx);
    (void) hipFree(d_iq);
    if (T.use_files) {                         /* the barrier file_cleanup relies on (the root reads the payloads inside t_gather_root, before it gets here) */
        void **fin = calloc((size_t) world, sizeof(*fin));
        uint64_t *fsz = calloc((size_t) world, sizeof(*fsz));
        if (fin && fsz && t_allgather(&T, NULL, 0, fin, fsz) == 0) { for (int r = 0; r < world; ++r) free(fin[r]); file_cleanup(&T); }
        free(fin); free(fsz);
    }
    if (!T.use_files) ncclCommDestroy(T.comm);
    (void) hipStreamDestroy(T.s);
    if (n) munmap((void *) iq, (size_t) st.st_size);
    close(fd);
    return rc;
}
