/* readsb_gpu_gather — the aggregator role in C, over RCCL (BASELINE.json configs[3]: independent sample streams, one per GPU,
 * "RCCL over xGMI used only to gather decoded-message lists/counts", host stays C).
 *
 * N processes, one per GPU and sample file.  Every rank demodulates its file on its GPU through the C ABI (modes_gpu.h); then
 *   1. an 8-byte ncclAllGather of the message counts,
 *   2. the 64-byte `struct mgpu_msg` records of every rank -> rank 0's HBM (one grouped ncclSend / ncclRecv exchange),
 *   3. rank 0 encodes the gathered records into one beast stream on its GPU (mgpu_beast_encode_device) — per rank in rank
 *      order, each rank's messages in stream order — and writes it to --out (what modesSendBeastOutput would have sent had the
 *      N receivers forwarded to one readsb over TCP, net_io.c:1655-1714; its ingest side is decodeBinMessage, net_io.c:3804-3959).
 * The same exchange as readsb_amd/gather.py (torch.distributed) for a host without Python.
 *
 * Bootstrap without MPI: rank 0 writes the ncclUniqueId to --id-file (temporary name + rename), the others wait for the file.
 *
 *   readsb_gpu_gather --rank R --world N --id-file /dev/shm/id --ifile stream_R.iq [--iformat UC8|SC16|SC16Q11]
 *                     [--fix|--no-fix|--aggressive] [--no-fix-df] [--preamble-threshold T] [--startup-time-ms T]
 *                     [--gpu-device D (default: rank)] [--out beast.bin (rank 0)]
 *                     [--forward-only [--net-rule] [--deferred-out deferred.bin]]
 *
 * --forward-only (round 6): rank 0 writes what N reference receivers would have FORWARDED, not every accepted
 * message — per rank's record list (a receiver has its own tracker: the aircraft table is reset between ranks) field decode ->
 * tracking gate -> gated beast encoder, all on the GPU (mgpu_decode_fields_device, mgpu_track_gate_device,
 * mgpu_beast_encode_gated_device): first messages of an aircraft are suppressed as outputMessage does (net_io.c:5846-5849),
 * --net-rule adds the network outputs' correctedbits < 2 (:5863-5872).  The few messages only a position tracker can settle are
 * left out of the stream and listed — {rank, index in the rank's list, offset in the stream} as three little-endian u64 each — in
 * --deferred-out (their count goes to stderr either way).
 */
#define __HIP_PLATFORM_AMD__ 1
#include <fcntl.h>
#include <inttypes.h>
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

static int exchange_id(const char *path, int rank, ncclUniqueId *id) {
    if (rank == 0) {
        if (ncclGetUniqueId(id) != ncclSuccess) return -1;
        char tmp[4096];
        snprintf(tmp, sizeof(tmp), "%s.tmp", path);
        FILE *f = fopen(tmp, "wb");
        if (!f || fwrite(id, sizeof(*id), 1, f) != 1) { perror(tmp); return -1; }
        fclose(f);
        return rename(tmp, path);
    }
    for (int tries = 0; tries < 60000; ++tries) {              /* up to a minute */
        FILE *f = fopen(path, "rb");
        if (f) {
            const size_t got = fread(id, sizeof(*id), 1, f);
            fclose(f);
            if (got == 1) return 0;
        }
        struct timespec ts = {0, 1000000};
        nanosleep(&ts, NULL);
    }
    fprintf(stderr, "rank %d: no ncclUniqueId in %s\n", rank, path);
    return -1;
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
    const char *ifile = NULL, *idfile = NULL, *outpath = NULL, *defpath = NULL;
    int rank = 0, world = 1, device = -1, forward_only = 0, net_rule = 0;
    const unsigned chunk_buffers = 512;
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
        else if (!strcmp(argv[i], "--out") && i + 1 < argc) outpath = argv[++i];
        else if (!strcmp(argv[i], "--forward-only")) forward_only = 1;
        else if (!strcmp(argv[i], "--net-rule")) net_rule = 1;
        else if (!strcmp(argv[i], "--deferred-out") && i + 1 < argc) defpath = argv[++i];
        else { fprintf(stderr, "unknown option %s\n", argv[i]); return 2; }
    }
    if (!ifile || !idfile || world < 1 || rank < 0 || rank >= world) {
        fprintf(stderr, "usage: %s --rank R --world N --id-file PATH --ifile FILE [--iformat F] [--fix|--no-fix|--aggressive] [--out beast.bin]\n", argv[0]);
        return 2;
    }
    cfg.device = device >= 0 ? device : rank;
    CHK_HIP(hipSetDevice(cfg.device));

    /* ---- this rank's stream: the whole file through mgpu_feed_iq, chunk by chunk ---- */
    const int fd = open(ifile, O_RDONLY);
    if (fd < 0) { perror(ifile); return 1; }
    struct stat st;
    if (fstat(fd, &st) != 0) { perror(ifile); return 1; }
    const size_t bps = cfg.format == MGPU_FMT_UC8 ? 2 : 4;
    const uint64_t nsamples = (uint64_t) st.st_size / bps;
    const uint8_t *iq = nsamples ? mmap(NULL, (size_t) st.st_size, PROT_READ, MAP_PRIVATE, fd, 0) : NULL;
    if (nsamples && iq == MAP_FAILED) { perror("mmap"); return 1; }
    cfg.max_samples = (uint64_t) chunk_buffers * cfg.buf_samples;
    mgpu_ctx *ctx = NULL;
    CHK_MGPU(mgpu_create(&cfg, &ctx), ctx);
    uint64_t cap = nsamples / 256 + 4096, nmsg = 0;
    struct mgpu_msg *msgs = malloc(cap * sizeof(*msgs));
    if (!msgs) return 1;
    for (uint64_t off = 0; off < nsamples; off += cfg.max_samples) {
        const uint64_t len = nsamples - off < cfg.max_samples ? nsamples - off : cfg.max_samples;
        CHK_MGPU(mgpu_feed_iq(ctx, iq + off * bps, len), ctx);
        for (;;) {
            const uint64_t pending = mgpu_pending_messages(ctx);
            if (!pending) break;
            if (nmsg + pending > cap) {
                cap = (nmsg + pending) * 2;
                msgs = realloc(msgs, cap * sizeof(*msgs));
                if (!msgs) return 1;
            }
            uint64_t got = 0;
            CHK_MGPU(mgpu_collect(ctx, msgs + nmsg, cap - nmsg, &got, NULL), ctx);
            nmsg += got;
        }
    }
    CHK_MGPU(mgpu_finish(ctx), ctx);

    /* ---- the exchange ---- */
    ncclUniqueId id;
    if (exchange_id(idfile, rank, &id) != 0) return 1;
    ncclComm_t comm;
    CHK_NCCL(ncclCommInitRank(&comm, world, id, rank));
    hipStream_t s;
    CHK_HIP(own_queue_stream(&s));
    unsigned long long *d_counts = NULL, mine = nmsg;
    CHK_HIP(hipMalloc((void **) &d_counts, (size_t) (world + 1) * sizeof(*d_counts)));
    CHK_HIP(hipMemcpyAsync(d_counts + world, &mine, sizeof(mine), hipMemcpyHostToDevice, s));
    CHK_NCCL(ncclAllGather(d_counts + world, d_counts, 1, ncclUint64, comm, s));
    unsigned long long *counts = malloc((size_t) world * sizeof(*counts));
    CHK_HIP(hipMemcpyAsync(counts, d_counts, (size_t) world * sizeof(*counts), hipMemcpyDeviceToHost, s));
    CHK_HIP(hipStreamSynchronize(s));
    uint64_t total = 0;
    for (int r = 0; r < world; ++r) total += counts[r];

    struct mgpu_msg *d_mine = NULL, *d_all = NULL;
    CHK_HIP(hipMalloc((void **) &d_mine, (nmsg + 1) * sizeof(*d_mine)));
    CHK_HIP(hipMemcpyAsync(d_mine, msgs, nmsg * sizeof(*d_mine), hipMemcpyHostToDevice, s));
    if (rank == 0) CHK_HIP(hipMalloc((void **) &d_all, (total + 1) * sizeof(*d_all)));
    CHK_NCCL(ncclGroupStart());
    if (rank == 0) {
        uint64_t off = 0;
        for (int r = 0; r < world; ++r) {
            if (counts[r]) CHK_NCCL(ncclRecv(d_all + off, (size_t) counts[r] * sizeof(*d_all), ncclUint8, r, comm, s));
            off += counts[r];
        }
    }
    if (nmsg) CHK_NCCL(ncclSend(d_mine, (size_t) nmsg * sizeof(*d_mine), ncclUint8, 0, comm, s));
    CHK_NCCL(ncclGroupEnd());
    CHK_HIP(hipStreamSynchronize(s));

    /* ---- rank 0: one beast stream of everything, encoded where the records are ---- */
    int rc = 0;
    if (rank == 0) {
        uint8_t *d_out = NULL;
        const uint64_t out_cap = total * 48 + 64;                  /* a frame is at most 2 + 2*(6 + 1 + 14) bytes */
        CHK_HIP(hipMalloc((void **) &d_out, out_cap));
        uint64_t bytes = 0, ndeferred_total = 0;
        if (!forward_only) {
            CHK_MGPU(mgpu_beast_encode_device(ctx, d_all, total, d_out, out_cap, &bytes), ctx);
        } else {
            struct mgpu_fields *d_fields = NULL;
            uint8_t *d_verdict = NULL;
            struct mgpu_deferred *d_def = NULL, *h_def = malloc((size_t) (total + 1) * sizeof(*h_def));
            CHK_HIP(hipMalloc((void **) &d_fields, (size_t) (total + 1) * sizeof(*d_fields)));
            CHK_HIP(hipMalloc((void **) &d_verdict, (size_t) total + 1));
            CHK_HIP(hipMalloc((void **) &d_def, (size_t) (total + 1) * sizeof(*d_def)));
            FILE *fd_def = defpath ? fopen(defpath, "wb") : NULL;
            if (defpath && !fd_def) { perror(defpath); return 1; }
            uint64_t off = 0;
            for (int r = 0; r < world; ++r) {                      /* every receiver has its own tracker */
                const uint64_t n = counts[r];
                if (n) {
                    uint64_t nb = 0, nd = 0;
                    CHK_MGPU(mgpu_track_gate_reset(ctx), ctx);
                    CHK_MGPU(mgpu_decode_fields_device(ctx, d_all + off, n, d_fields), ctx);
                    CHK_MGPU(mgpu_track_gate_device(ctx, d_all + off, d_fields, n, d_verdict), ctx);
                    CHK_MGPU(mgpu_beast_encode_gated_device(ctx, d_all + off, d_verdict, n, net_rule ? MGPU_BEAST_NET_RULE : 0u, d_out + bytes,
                                                            out_cap - bytes, &nb, d_def, n, &nd), ctx);
                    if (nd && fd_def) {
                        CHK_HIP(hipMemcpy(h_def, d_def, (size_t) nd * sizeof(*h_def), hipMemcpyDeviceToHost));
                        for (uint64_t k = 0; k < nd; ++k) {
                            const uint64_t rec[3] = {(uint64_t) r, h_def[k].index, bytes + h_def[k].offset};
                            if (fwrite(rec, sizeof(rec), 1, fd_def) != 1) rc = 1;
                        }
                    }
                    bytes += nb;
                    ndeferred_total += nd;
                }
                off += n;
            }
            if (fd_def) fclose(fd_def);
            free(h_def);
            (void) hipFree(d_def); (void) hipFree(d_verdict); (void) hipFree(d_fields);
        }
        uint8_t *out = malloc(bytes + 1);
        CHK_HIP(hipMemcpy(out, d_out, bytes, hipMemcpyDeviceToHost));
        FILE *f = outpath ? fopen(outpath, "wb") : stdout;
        if (!f) { perror(outpath); return 1; }
        if (bytes && fwrite(out, 1, bytes, f) != bytes) rc = 1;
        if (outpath) fclose(f); else fflush(stdout);
        fprintf(stderr, "readsb_gpu_gather: %d rank(s), %" PRIu64 " messages gathered (", world, total);
        for (int r = 0; r < world; ++r) fprintf(stderr, "%s%llu", r ? " + " : "", counts[r]);
        fprintf(stderr, "), %" PRIu64 " beast bytes", bytes);
        if (forward_only) fprintf(stderr, " (forwarded frames only; %" PRIu64 " message(s) left to a position tracker)", ndeferred_total);
        fprintf(stderr, "\n");
        free(out);
        (void) hipFree(d_out);
        (void) hipFree(d_all);
    }
    (void) hipFree(d_mine);
    (void) hipFree(d_counts);
    free(counts);
    free(msgs);
    ncclCommDestroy(comm);
    (void) hipStreamDestroy(s);
    mgpu_destroy(ctx);
    if (nsamples) munmap((void *) iq, (size_t) st.st_size);
    close(fd);
    return rc;
}
