/* synth_iq.c — seeded synthetic 1090 MHz IQ capture generator (2.4 MSps).
 *
 * There are no recorded IQ fixtures in the reference tree, so every
 * parity test and the benchmark run on captures produced here.  The recipe
 * follows SURVEY §8(d): a 12 MHz-tick pulse envelope (preamble pulses at 0, 1.0,
 * 3.5, 4.5 us, PPM data bits from 8 us, 0.5 us pulses), box-averaged over the 5
 * ticks of each 2.4 MSps sample, placed on a random carrier phase, plus uniform
 * noise, quantised to UC8 / SC16 / SC16Q11.
 *
 * Deterministic by construction: the stream is generated in blocks of
 * SYNTH_BLOCK samples, the message schedule and the noise of block b depend only
 * on (seed, b), so any number of threads and any sub-range give identical bytes.
 *
 * Traffic mix (per message):  55 % DF17 (valid PI), 2 % DF18, 13 % DF11
 * (PI = CRC ^ IID, IID 0 two thirds of the time), 15 % DF4/5, 10 % DF20/21,
 * 5 % DF0/16 (AP = CRC ^ ICAO); 8 % of frames get one random bit flipped and 3 %
 * a second one (exercises --fix / --aggressive / DF repair).  ICAOs come from a
 * pool of `naircraft` addresses 0x400000+r; aircraft r stops squittering
 * (DF17/11) for 150 s out of every 450 s while still replying (DF4/5/20/21) so
 * the ICAO-filter expiry path is exercised on long captures; 3 % of address/
 * parity replies come from aircraft that never squitter.
 *
 * Build:  gcc -O2 -shared -fPIC -o libsynth_iq.so synth_iq.c -lm -lpthread
 *         gcc -O2 -DSYNTH_MAIN -o synth_iq synth_iq.c -lm -lpthread
 */
#define _GNU_SOURCE
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define SYNTH_BLOCK 2400000ULL  /* samples per generation block = 1 s */
#define SYNTH_FMT_UC8 0
#define SYNTH_FMT_SC16 1
#define SYNTH_FMT_SC16Q11 2
#define MSG_TICKS ((8 + 112) * 12)  /* longest frame in 12 MHz ticks */

struct synth_cfg {
    uint64_t seed;
    int format;          /* SYNTH_FMT_* */
    double msgs_per_sec; /* mean frame rate */
    int naircraft;       /* ICAO pool size */
    int dense;           /* bit 0: DF17-only bursts (config 5: overlapping 112-bit frames); bit 1: add Mode A/C replies;
                          * bit 2: Gaussian instead of uniform noise (sigma = noise_lsb) */
    double noise_lsb;    /* uniform noise amplitude, +-noise_lsb LSB of UC8 */
};

static inline uint64_t splitmix64(uint64_t *s) {
    uint64_t z = (*s += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
static inline uint64_t xs64(uint64_t *s) {
    uint64_t x = *s;
    x ^= x << 13; x ^= x >> 7; x ^= x << 17;
    return *s = x;
}
static inline double urand(uint64_t *s) { return (xs64(s) >> 11) * (1.0 / 9007199254740992.0); }
static inline uint32_t irand(uint64_t *s, uint32_t n) { return (uint32_t) ((xs64(s) >> 16) % n); }

/* Mode S CRC-24, generator 0xFFF409, bitwise (independent of the product's tables) */
static uint32_t crc24(const uint8_t *msg, int nbytes_data) {
    uint32_t rem = 0;
    for (int i = 0; i < nbytes_data; i++) {
        rem ^= (uint32_t) msg[i] << 16;
        for (int j = 0; j < 8; j++)
            rem = (rem & 0x800000) ? ((rem << 1) ^ 0xFFF409) & 0xFFFFFF : (rem << 1) & 0xFFFFFF;
    }
    return rem;
}

struct frame {
    int64_t t0;       /* start tick (12 MHz) of the preamble, absolute */
    int nbits;
    double aI, aQ;    /* amplitude * cos/sin(carrier phase), UC8 LSB */
    uint8_t msg[14];
};

static void make_frame(const struct synth_cfg *cfg, uint64_t *rng, int64_t block, struct frame *f) {
    int64_t block_t0 = block * (int64_t) SYNTH_BLOCK * 5;
    f->t0 = block_t0 + (int64_t) (urand(rng) * SYNTH_BLOCK * 5);
    double amp = 40.0 + urand(rng) * 80.0;
    if (irand(rng, 100) < 2) amp = 120.0 + urand(rng) * 7.0; /* a few > -3 dBFS */
    double ph = urand(rng) * 6.283185307179586;
    f->aI = amp * cos(ph);
    f->aQ = amp * sin(ph);
    int64_t sec = block;  /* 1 block == 1 s */
    uint32_t r = irand(rng, cfg->naircraft);
    uint32_t icao = 0x400000 + r;
    int silent = ((sec + 50 * (int64_t) r) / 150) % 3 == 0; /* squitter pause */
    uint32_t kind = irand(rng, 100);
    uint8_t *m = f->msg;
    memset(m, 0, 14);
    if (cfg->dense & 1) kind = 0;
    if (kind < 70 && silent && !(cfg->dense & 1)) kind = 70 + irand(rng, 30); /* replies only */
    if (kind >= 70 && irand(rng, 100) < 3) icao = 0xA00000 + irand(rng, 50); /* never squitters */
    if (kind < 57) {            /* DF17 (55 %) / DF18 (2 %) */
        m[0] = (kind < 55 ? (17 << 3) : (18 << 3)) | (kind < 55 ? 5 : irand(rng, 7));
        m[1] = icao >> 16; m[2] = icao >> 8; m[3] = icao;
        for (int i = 4; i < 11; i++) m[i] = xs64(rng) >> 24;
        uint32_t c = crc24(m, 11);
        m[11] = c >> 16; m[12] = c >> 8; m[13] = c;
        f->nbits = 112;
    } else if (kind < 70) {     /* DF11 */
        m[0] = (11 << 3) | 5;
        m[1] = icao >> 16; m[2] = icao >> 8; m[3] = icao;
        uint32_t iid = irand(rng, 3) ? 0 : 1 + irand(rng, 63);
        uint32_t c = crc24(m, 4) ^ iid;
        m[4] = c >> 16; m[5] = c >> 8; m[6] = c;
        f->nbits = 56;
    } else if (kind < 85) {     /* DF4 / DF5 */
        m[0] = ((kind & 1 ? 4 : 5) << 3) | irand(rng, 6);
        for (int i = 1; i < 4; i++) m[i] = xs64(rng) >> 24;
        uint32_t c = crc24(m, 4) ^ icao;
        m[4] = c >> 16; m[5] = c >> 8; m[6] = c;
        f->nbits = 56;
    } else if (kind < 95) {     /* DF20 / DF21 */
        m[0] = ((kind & 1 ? 20 : 21) << 3) | irand(rng, 6);
        for (int i = 1; i < 11; i++) m[i] = xs64(rng) >> 24;
        uint32_t c = crc24(m, 11) ^ icao;
        m[11] = c >> 16; m[12] = c >> 8; m[13] = c;
        f->nbits = 112;
    } else if (kind < 98) {     /* DF0 */
        m[0] = (0 << 3) | irand(rng, 8);
        for (int i = 1; i < 4; i++) m[i] = xs64(rng) >> 24;
        uint32_t c = crc24(m, 4) ^ icao;
        m[4] = c >> 16; m[5] = c >> 8; m[6] = c;
        f->nbits = 56;
    } else {                    /* DF16 */
        m[0] = (16 << 3) | irand(rng, 8);
        for (int i = 1; i < 11; i++) m[i] = xs64(rng) >> 24;
        uint32_t c = crc24(m, 11) ^ icao;
        m[11] = c >> 16; m[12] = c >> 8; m[13] = c;
        f->nbits = 112;
    }
    if (irand(rng, 100) < 8) { uint32_t b = irand(rng, f->nbits); m[b >> 3] ^= 0x80 >> (b & 7); }
    if (irand(rng, 100) < 3) { uint32_t b = irand(rng, f->nbits); m[b >> 3] ^= 0x80 >> (b & 7); }
}

/* frames of block b: count ~ Poisson-ish (fixed fraction jitter), all from (seed,b) */
static int block_frames(const struct synth_cfg *cfg, int64_t block, struct frame **out) {
    if (block < 0) { *out = NULL; return 0; }
    uint64_t s = cfg->seed ^ (0xD1B54A32D192ED03ULL * (uint64_t) (block + 1));
    uint64_t rng = splitmix64(&s) | 1;
    int n = (int) (cfg->msgs_per_sec * (0.9 + 0.2 * urand(&rng)) + 0.5);
    struct frame *f = malloc(sizeof(*f) * (n ? n : 1));
    for (int i = 0; i < n; i++) make_frame(cfg, &rng, block, &f[i]);
    *out = f;
    return n;
}

static inline void add_pulse_w(float *accI, float *accQ, int64_t s_lo, int64_t s_hi, int64_t p, int width, double aI, double aQ) {
    /* pulse occupies ticks [p, p+width); sample s covers ticks [5s, 5s+5) */
    int64_t s0 = p / 5, s1 = (p + width - 1) / 5;
    for (int64_t s = s0; s <= s1; s++) {
        if (s < s_lo || s >= s_hi) continue;
        int64_t lo = s * 5 > p ? s * 5 : p;
        int64_t hi = s * 5 + 5 < p + width ? s * 5 + 5 : p + width;
        if (hi <= lo) continue;
        float w = (float) (hi - lo) * 0.2f;
        accI[s - s_lo] += (float) aI * w;
        accQ[s - s_lo] += (float) aQ * w;
    }
}

static inline void add_pulse(float *accI, float *accQ, int64_t s_lo, int64_t s_hi, int64_t p, double aI, double aQ) {
    add_pulse_w(accI, accQ, s_lo, s_hi, p, 6, aI, aQ);
}

/* Mode A/C replies (bit 1 of `dense`): 0.45 us pulses on a 1.45 us raster — F1, C1 A1 C2 A2 C4 A4, X, B1 D1 B2 D2 B4 D4,
 * F2, two empty slots, SPI (demod_2400.c:588-616) — from their own random stream, so that the Mode S traffic of a
 * seed is the same with and without them.  On the 12 MHz tick grid: 5-tick pulses at round(17.4 k). */
struct acreply { int64_t t0; double aI, aQ; uint32_t bits; };

static int block_acreplies(const struct synth_cfg *cfg, int64_t block, struct acreply **out) {
    if (block < 0 || !(cfg->dense & 2)) { *out = NULL; return 0; }
    uint64_t s = cfg->seed ^ (0x9E3779B97F4A7C15ULL * (uint64_t) (block + 7));
    uint64_t rng = splitmix64(&s) | 1;
    int n = (int) (cfg->msgs_per_sec * (0.9 + 0.2 * urand(&rng)) + 0.5);      /* as many replies as Mode S frames */
    struct acreply *r = malloc(sizeof(*r) * (n ? n : 1));
    for (int i = 0; i < n; i++) {
        r[i].t0 = block * (int64_t) SYNTH_BLOCK * 5 + (int64_t) (urand(&rng) * SYNTH_BLOCK * 5);
        double amp = 50.0 + urand(&rng) * 70.0, ph = urand(&rng) * 6.283185307179586;
        r[i].aI = amp * cos(ph); r[i].aQ = amp * sin(ph);
        uint32_t code = irand(&rng, 4096);                 /* C1 A1 C2 A2 C4 A4 | B1 D1 B2 D2 B4 D4 */
        uint32_t bits = (1u << 19) | (1u << 5);            /* F1 = bit 0 (MSB of 20), F2 = bit 14 */
        bits |= ((code >> 6) & 0x3f) << 13;                /* slots 1..6 */
        bits |= (code & 0x3f) << 6;                        /* slots 8..13 */
        if (irand(&rng, 100) < 5) bits |= 1u << 2;         /* SPI, slot 17 */
        r[i].bits = bits;
    }
    *out = r;
    return n;
}

static void render_acreply(const struct acreply *r, float *accI, float *accQ, int64_t s_lo, int64_t s_hi) {
    if ((r->t0 + 20 * 18) / 5 + 2 < s_lo || r->t0 / 5 > s_hi) return;
    for (int k = 0; k < 20; k++)
        if ((r->bits >> (19 - k)) & 1) add_pulse_w(accI, accQ, s_lo, s_hi, r->t0 + (int64_t) (17.4 * k + 0.5), 5, r->aI, r->aQ);
}

static void render_frame(const struct frame *f, float *accI, float *accQ, int64_t s_lo, int64_t s_hi) {
    if ((f->t0 + MSG_TICKS) / 5 + 2 < s_lo || f->t0 / 5 > s_hi) return;
    static const int pre[4] = {0, 12, 42, 54};
    for (int i = 0; i < 4; i++) add_pulse(accI, accQ, s_lo, s_hi, f->t0 + pre[i], f->aI, f->aQ);
    for (int k = 0; k < f->nbits; k++) {
        int bit = (f->msg[k >> 3] >> (7 - (k & 7))) & 1;
        add_pulse(accI, accQ, s_lo, s_hi, f->t0 + 96 + 12 * k + (bit ? 0 : 6), f->aI, f->aQ);
    }
}

/* generate samples [first, first+count) of the stream into out (format bytes) */
static void gen_range(const struct synth_cfg *cfg, uint64_t first, uint64_t count, uint8_t *out) {
    const int bps = cfg->format == SYNTH_FMT_UC8 ? 2 : 4;
    uint64_t pos = first, end = first + count;
    float *accI = malloc(sizeof(float) * SYNTH_BLOCK), *accQ = malloc(sizeof(float) * SYNTH_BLOCK);
    while (pos < end) {
        int64_t b = pos / SYNTH_BLOCK;
        uint64_t b_lo = b * SYNTH_BLOCK, b_hi = b_lo + SYNTH_BLOCK;
        uint64_t lo = pos, hi = end < b_hi ? end : b_hi;
        memset(accI, 0, sizeof(float) * SYNTH_BLOCK);
        memset(accQ, 0, sizeof(float) * SYNTH_BLOCK);
        for (int64_t bb = b - 1; bb <= b; bb++) {
            struct frame *fr; int n = block_frames(cfg, bb, &fr);
            for (int i = 0; i < n; i++) render_frame(&fr[i], accI, accQ, (int64_t) b_lo, (int64_t) b_hi);
            free(fr);
            struct acreply *ar; int na = block_acreplies(cfg, bb, &ar);
            for (int i = 0; i < na; i++) render_acreply(&ar[i], accI, accQ, (int64_t) b_lo, (int64_t) b_hi);
            free(ar);
        }
        /* noise: per-block stream, skipped ahead to `lo` so sub-ranges are reproducible */
        uint64_t s = cfg->seed ^ (0xA0761D6478BD642FULL * (uint64_t) (b + 1));
        uint64_t rng = splitmix64(&s) | 1;
        for (uint64_t i = b_lo; i < lo; i++) { xs64(&rng); xs64(&rng); }
        for (uint64_t i = lo; i < hi; i++) {
            double u1 = (xs64(&rng) >> 11) * (1.0 / 9007199254740992.0), u2 = (xs64(&rng) >> 11) * (1.0 / 9007199254740992.0);
            double nI, nQ;
            if (cfg->dense & 4) {        /* Gaussian I/Q noise, sigma = noise_lsb (Rayleigh magnitudes): Box-Muller on the same two draws */
                const double rr = sqrt(-2.0 * log(u1 > 1e-300 ? u1 : 1e-300)) * cfg->noise_lsb;
                nI = rr * cos(6.283185307179586 * u2);
                nQ = rr * sin(6.283185307179586 * u2);
            } else {
                nI = (u1 * 2.0 - 1.0) * cfg->noise_lsb;
                nQ = (u2 * 2.0 - 1.0) * cfg->noise_lsb;
            }
            double vI = accI[i - b_lo] + nI, vQ = accQ[i - b_lo] + nQ;  /* centred, UC8 LSB */
            uint8_t *o = out + (i - first) * bps;
            if (cfg->format == SYNTH_FMT_UC8) {
                double a = floor(vI + 127.5 + 0.5), c = floor(vQ + 127.5 + 0.5);
                o[0] = a < 0 ? 0 : a > 255 ? 255 : (uint8_t) a;
                o[1] = c < 0 ? 0 : c > 255 ? 255 : (uint8_t) c;
            } else {
                double sc = cfg->format == SYNTH_FMT_SC16Q11 ? 16.0 : 256.0;
                double lim = cfg->format == SYNTH_FMT_SC16Q11 ? 2047.0 : 32767.0;
                double a = floor(vI * sc + 0.5), c = floor(vQ * sc + 0.5);
                a = a < -lim ? -lim : a > lim ? lim : a;
                c = c < -lim ? -lim : c > lim ? lim : c;
                int16_t ia = (int16_t) a, ic = (int16_t) c;
                o[0] = ia & 0xff; o[1] = (ia >> 8) & 0xff; o[2] = ic & 0xff; o[3] = (ic >> 8) & 0xff;
            }
        }
        pos = hi;
    }
    free(accI); free(accQ);
}

struct job { const struct synth_cfg *cfg; uint64_t first, count; uint8_t *out; };
static void *job_main(void *p) { struct job *j = p; gen_range(j->cfg, j->first, j->count, j->out); return NULL; }

/* Public entry: fill `out` with samples [first, first+count) using up to nthreads threads. */
int synth_iq_generate(uint64_t seed, int format, double msgs_per_sec, int naircraft, int dense,
                      double noise_lsb, uint64_t first, uint64_t count, void *out, int nthreads) {
    struct synth_cfg cfg = {seed, format, msgs_per_sec, naircraft > 0 ? naircraft : 200, dense, noise_lsb};
    const int bps = format == SYNTH_FMT_UC8 ? 2 : 4;
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    /* split on block boundaries */
    uint64_t nblocks = (first + count + SYNTH_BLOCK - 1) / SYNTH_BLOCK - first / SYNTH_BLOCK;
    if ((uint64_t) nthreads > nblocks) nthreads = (int) nblocks;
    if (nthreads <= 1) { gen_range(&cfg, first, count, out); return 0; }
    pthread_t th[256]; struct job jobs[256];
    uint64_t b0 = first / SYNTH_BLOCK;
    for (int t = 0; t < nthreads; t++) {
        uint64_t blo = b0 + nblocks * t / nthreads, bhi = b0 + nblocks * (t + 1) / nthreads;
        uint64_t lo = blo * SYNTH_BLOCK, hi = bhi * SYNTH_BLOCK;
        if (lo < first) lo = first;
        if (hi > first + count) hi = first + count;
        jobs[t] = (struct job){&cfg, lo, hi - lo, (uint8_t *) out + (lo - first) * bps};
        pthread_create(&th[t], NULL, job_main, &jobs[t]);
    }
    for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
    return 0;
}

#ifdef SYNTH_MAIN
/* synth_iq <out.iq> <UC8|SC16|SC16Q11> <seconds> [seed] [msgs_per_sec] [naircraft] [dense] [threads] */
int main(int argc, char **argv) {
    if (argc < 4) {
        fprintf(stderr, "usage: %s <out.iq> <UC8|SC16|SC16Q11> <seconds> [seed] [msgs/s] [naircraft] [dense] [threads]\n", argv[0]);
        return 2;
    }
    int format = !strcmp(argv[2], "UC8") ? 0 : !strcmp(argv[2], "SC16") ? 1 : 2;
    uint64_t n = (uint64_t) (atof(argv[3]) * 2400000.0);
    uint64_t seed = argc > 4 ? strtoull(argv[4], NULL, 0) : 88172645463325252ULL;
    double rate = argc > 5 ? atof(argv[5]) : 2000.0;
    int nac = argc > 6 ? atoi(argv[6]) : 200;
    int dense = argc > 7 ? atoi(argv[7]) : 0;
    int threads = argc > 8 ? atoi(argv[8]) : 8;
    size_t bytes = n * (format == 0 ? 2 : 4);
    void *buf = malloc(bytes ? bytes : 1);
    synth_iq_generate(seed, format, rate, nac, dense, 3.0, 0, n, buf, threads);
    FILE *f = fopen(argv[1], "wb");
    if (!f) { perror(argv[1]); return 1; }
    fwrite(buf, 1, bytes, f);
    fclose(f);
    return 0;
}
#endif
