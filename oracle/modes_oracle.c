/* modes_oracle.c — plain-C CPU restatement of readsb's Mode-S demodulator hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see modes_oracle.h).  Parity PINNED against oracle/_ref
 * (the reference's own objects) by tests/test_oracle.py and tests/golden/.
 *
 * Every function cites the reference code it restates (file:line under
 * /root/reference).  The restatement is sequential and deliberately simple; it is
 * the specification the HIP kernels in readsb_amd/csrc are checked against.
 */
#ifndef _GNU_SOURCE
#define _GNU_SOURCE
#endif
#include "modes_oracle.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define BUF_SAMPLES 131072u   /* Modes.sdr_buf_samples, readsb.c:228,2212 */
#define TRAILING 326u         /* (8+112+16)*2.4, readsb.c:288 */
#define FILTER_TTL_MS 60000   /* MODES_ICAO_FILTER_TTL, readsb.h:315 */
#define BADDR 0xff123456u     /* Modes.show_only default, readsb.h:296 */

/* ------------------------------------------------------------------ convert.c */

static uint16_t *g_uc8_lut;

/* init_uc8_lookup, convert.c:35-62.  Index = the little-endian u16 formed by the
 * (I,Q) byte pair = I | Q<<8; the table is symmetric in (I,Q). */
const uint16_t *modes_oracle_uc8_lut(void) {
    if (g_uc8_lut) return g_uc8_lut;
    g_uc8_lut = malloc(65536 * sizeof(uint16_t));
    for (int i = 0; i <= 255; i++) {
        for (int q = 0; q <= 255; q++) {
            float fI = (i - 127.5) / 127.5;   /* double expression rounded to float */
            float fQ = (q - 127.5) / 127.5;
            float magsq = fI * fI + fQ * fQ;
            if (magsq > 1) magsq = 1;
            float mag = sqrtf(magsq);
            g_uc8_lut[i * 256 + q] = (uint16_t) (mag * 65535.0f + 0.5f);
        }
    }
    return g_uc8_lut;
}

/* convert_uc8_nodc convert.c:64-108; convert_sc16_nodc :212-250; convert_sc16q11_nodc :329-367 */
void modes_oracle_convert(int format, const uint8_t *iq, uint16_t *mag, uint64_t n,
                          double *mean_level, double *mean_power) {
    if (format == ORACLE_FMT_UC8) {
        const uint16_t *lut = modes_oracle_uc8_lut();
        uint64_t sum_level = 0, sum_power = 0;
        for (uint64_t i = 0; i < n; i++) {
            uint16_t m = lut[iq[2 * i] | (iq[2 * i + 1] << 8)];
            mag[i] = m;
            sum_level += m;
            sum_power += (uint32_t) m * (uint32_t) m;
        }
        if (mean_level) *mean_level = sum_level / 65536.0 / n;          /* sic: 65536 */
        if (mean_power) *mean_power = sum_power / 65535.0 / 65535.0 / n;
        return;
    }
    const float scale = format == ORACLE_FMT_SC16 ? 32768.0f : 2048.0f;
    float sum_level = 0, sum_power = 0;     /* float running sums, order dependent */
    for (uint64_t i = 0; i < n; i++) {
        int16_t I = (int16_t) (iq[4 * i] | (iq[4 * i + 1] << 8));
        int16_t Q = (int16_t) (iq[4 * i + 2] | (iq[4 * i + 3] << 8));
        float fI = I / scale, fQ = Q / scale;
        float magsq = fI * fI + fQ * fQ;
        if (magsq > 1) magsq = 1;
        float m = sqrtf(magsq);
        sum_power += magsq;
        sum_level += m;
        mag[i] = (uint16_t) (m * 65535.0f + 0.5f);
    }
    if (mean_level) *mean_level = sum_level / n;
    if (mean_power) *mean_power = sum_power / n;
}

/* ---------------------------------------------------------------------- crc.c */

static uint32_t crc_table[256];
static uint32_t single_bit_syndrome[112];
static int crc_ready;

struct einfo { uint32_t syndrome; int errors; int bit[2]; };
static struct einfo *tab_short, *tab_long;
static int n_short, n_long;

static void crc_lookup_init(void);

/* modesChecksum, crc.c:67-82 */
uint32_t modes_oracle_checksum(const uint8_t *msg, int bits) {
    if (!crc_ready) crc_lookup_init();
    uint32_t rem = 0;
    int n = bits / 8;
    for (int i = 0; i < n - 3; ++i) {
        rem = (rem << 8) ^ crc_table[msg[i] ^ ((rem & 0xff0000) >> 16)];
        rem &= 0xffffff;
    }
    return rem ^ (msg[n - 3] << 16) ^ (msg[n - 2] << 8) ^ msg[n - 1];
}

/* initLookupTables, crc.c:42-64 */
static void crc_lookup_init(void) {
    if (crc_ready) return;
    for (int i = 0; i < 256; ++i) {
        uint32_t c = i << 16;
        for (int j = 0; j < 8; ++j)
            c = (c & 0x800000) ? (c << 1) ^ 0xfff409U : (c << 1);
        crc_table[i] = c & 0xffffff;
    }
    crc_ready = 1;
    uint8_t msg[14];
    memset(msg, 0, sizeof(msg));
    for (int i = 0; i < 112; ++i) {
        msg[i / 8] ^= 1 << (7 - (i & 7));
        single_bit_syndrome[i] = modes_oracle_checksum(msg, 112);
        msg[i / 8] ^= 1 << (7 - (i & 7));
    }
}

static int einfo_cmp(const void *a, const void *b) {
    const struct einfo *x = a, *y = b;
    return (x->syndrome > y->syndrome) - (x->syndrome < y->syndrome);
}

static struct einfo *tab_find(struct einfo *t, int n, uint32_t syn) {
    int lo = 0, hi = n - 1;
    while (lo <= hi) {
        int mid = (lo + hi) / 2;
        if (t[mid].syndrome == syn) return &t[mid];
        if (t[mid].syndrome < syn) lo = mid + 1; else hi = mid - 1;
    }
    return NULL;
}

/* prepareErrorTable, crc.c:180-350: all 1..max_correct-bit error patterns over message bits
 * 5..bits-1 (the DF field is never corrected here), sorted by syndrome; syndromes produced by
 * more than one pattern are dropped entirely; then every entry whose syndrome is also produced
 * by a (max_correct+1 .. max_detect)-bit pattern is dropped (flagCollisions, crc.c:151-175). */
static struct einfo *build_table(int bits, int max_correct, int max_detect, int *size_out) {
    *size_out = 0;
    if (!max_correct) return NULL;
    const int offset = 112 - bits, nb = bits - 5;
    int maxsize = nb + (max_correct > 1 ? nb * (nb - 1) / 2 : 0);
    struct einfo *t = malloc(maxsize * sizeof(*t));
    int n = 0;
    for (int i = 5; i < bits; ++i) {
        t[n++] = (struct einfo){single_bit_syndrome[i + offset], 1, {i, -1}};
        if (max_correct > 1)
            for (int j = i + 1; j < bits; ++j)
                t[n++] = (struct einfo){single_bit_syndrome[i + offset] ^ single_bit_syndrome[j + offset], 2, {i, j}};
    }
    qsort(t, n, sizeof(*t), einfo_cmp);
    int m = 0;
    for (int i = 0; i < n;) {           /* drop ambiguous syndromes, crc.c:232-249 */
        int j = i;
        while (j + 1 < n && t[j + 1].syndrome == t[i].syndrome) ++j;
        if (j == i) t[m++] = t[i];
        i = j + 1;
    }
    n = m;
    if (max_detect > max_correct) {     /* crc.c:252-283 with first_error = max_correct+1 */
        for (int a = 5; a < bits; ++a) {
            uint32_t sa = single_bit_syndrome[a + offset];
            for (int b = a + 1; b < bits; ++b) {
                uint32_t sb = sa ^ single_bit_syndrome[b + offset];
                for (int c = b + 1; c < bits; ++c) {
                    uint32_t sc = sb ^ single_bit_syndrome[c + offset];
                    struct einfo *e;
                    if (max_correct < 3 && max_detect >= 3 && (e = tab_find(t, n, sc))) e->errors = -1;
                    if (max_detect >= 4)
                        for (int d = c + 1; d < bits; ++d) {
                            uint32_t sd = sc ^ single_bit_syndrome[d + offset];
                            if ((e = tab_find(t, n, sd))) e->errors = -1;
                        }
                }
            }
        }
        m = 0;
        for (int i = 0; i < n; ++i)
            if (t[i].errors != -1) t[m++] = t[i];
        n = m;
    }
    *size_out = n;
    return t;
}

/* modesChecksumInit, crc.c:353-378 */
void modes_oracle_crc_init(int nfix) {
    crc_lookup_init();
    free(tab_short); free(tab_long);
    tab_short = tab_long = NULL; n_short = n_long = 0;
    if (nfix == 1) {
        tab_short = build_table(56, 1, 1, &n_short);
        tab_long = build_table(112, 1, 1, &n_long);
    } else if (nfix >= 2) {
        tab_short = build_table(56, 2, 4, &n_short);
        tab_long = build_table(112, 2, 4, &n_long);
    }
}

int modes_oracle_table_size(int bits) { return bits == 56 ? n_short : n_long; }

static const struct einfo NO_ERRORS = {0, 0, {-1, -1}};

/* modesChecksumDiagnose, crc.c:383-406 */
static const struct einfo *diagnose(uint32_t syndrome, int bitlen) {
    if (syndrome == 0) return &NO_ERRORS;
    if (bitlen == 56) return tab_short ? tab_find(tab_short, n_short, syndrome) : NULL;
    return tab_long ? tab_find(tab_long, n_long, syndrome) : NULL;
}

int modes_oracle_diagnose(uint32_t syndrome, int bits, int *b0, int *b1) {
    const struct einfo *e = diagnose(syndrome, bits);
    if (!e) return -1;
    *b0 = e->errors > 0 ? e->bit[0] : -1;
    *b1 = e->errors > 1 ? e->bit[1] : -1;
    return e->errors;
}

/* modesChecksumFix, crc.c:410-418 */
static void checksum_fix(uint8_t *msg, const struct einfo *e) {
    for (int i = 0; i < e->errors; ++i)
        msg[e->bit[i] >> 3] ^= 1 << (7 - (e->bit[i] & 7));
}

/* -------------------------------------------------------------- icao_filter.c
 * Semantic model of the two-generation open-addressed table (icao_filter.c:29-154).
 * Bucket placement never influences results; what does: membership in the active and
 * inactive generation, `occupied` (new inserts into the active generation since the last
 * flip/resize) and the table size, because growing (occupied > buckets/3) re-inserts only the
 * ACTIVE generation and silently empties the inactive one (icaoFilterResize, :65-93). */
struct gen { uint8_t *present; uint32_t *members; uint32_t n, cap; int has_baddr; };
static struct gen g_gen[2];
static int g_active;
static uint32_t g_occupied, g_filter_bits;

static void gen_clear(struct gen *g) {
    for (uint32_t i = 0; i < g->n; i++) g->present[g->members[i]] = 0;
    g->n = 0; g->has_baddr = 0;
}
static int gen_has(const struct gen *g, uint32_t addr) {
    return addr < (1u << 24) ? g->present[addr] : (addr == BADDR && g->has_baddr);
}
static int gen_add(struct gen *g, uint32_t addr) {   /* returns 1 if newly inserted */
    if (gen_has(g, addr)) return 0;
    if (addr >= (1u << 24)) { g->has_baddr = 1; return 1; }
    if (g->n == g->cap) { g->cap = g->cap ? g->cap * 2 : 1024; g->members = realloc(g->members, g->cap * sizeof(uint32_t)); }
    g->present[addr] = 1;
    g->members[g->n++] = addr;
    return 1;
}
static uint32_t gen_size(const struct gen *g) { return g->n + (g->has_baddr ? 1 : 0); }

static void filter_init(void) {                       /* icaoFilterInit, :47-59 */
    for (int i = 0; i < 2; i++) {
        if (!g_gen[i].present) g_gen[i].present = calloc(1u << 24, 1);
        gen_clear(&g_gen[i]);
    }
    g_active = 0; g_occupied = 0; g_filter_bits = 8;
}
static void filter_resize(uint32_t bits) {            /* icaoFilterResize, :65-93 */
    g_filter_bits = bits;
    gen_clear(&g_gen[!g_active]);                     /* the inactive generation is lost */
    g_occupied = gen_size(&g_gen[g_active]);          /* re-add of the active generation */
}
static void filter_add(uint32_t addr) {               /* icaoFilterAdd, :112-130 */
    if (gen_add(&g_gen[g_active], addr)) g_occupied++;
    if (g_occupied > (1u << g_filter_bits) / 3 && g_filter_bits < 20)
        filter_resize(g_filter_bits + 1);
}
static int filter_test(uint32_t addr) {               /* icaoFilterTest, :132-154 */
    return gen_has(&g_gen[0], addr) || gen_has(&g_gen[1], addr);
}
static void filter_expire(void) {                     /* icaoFilterExpire, :96-110 */
    if (g_occupied < (1u << g_filter_bits) / 9 && g_filter_bits > 8)
        filter_resize(g_filter_bits - 1);
    g_occupied = 0;
    gen_clear(&g_gen[!g_active]);
    g_active = !g_active;
}

/* ------------------------------------------------------------------- mode_s.c */

static int g_nfix, g_fixDF;

/* getbits(msg, 9, 32), mode_s.h:56-113 */
static uint32_t aa_field(const uint8_t *msg) { return (msg[1] << 16) | (msg[2] << 8) | msg[3]; }

/* correct_aa_field, mode_s.c:230-245 */
static void correct_aa(uint32_t *addr, const struct einfo *e) {
    for (int i = 0; i < e->errors; ++i)
        if (e->bit[i] >= 8 && e->bit[i] <= 31) *addr ^= 1u << (31 - e->bit[i]);
}

/* fixDF17msgtype, mode_s.c:276-301: DF one bit away from 17 and CRC-112 clean once DF:=17 */
static int fix_df17(uint8_t *msg, int *msgtype) {
    if (!g_fixDF || !g_nfix) return 0;
    switch (*msgtype) {
        case 1: case 25: case 21: case 19: case 16: {
            uint8_t orig = msg[0];
            msg[0] = (msg[0] & 7) | (17 << 3);
            if (modes_oracle_checksum(msg, 112) == 0) { *msgtype = 17; return orig ? orig : 0; }
            msg[0] = orig;
            return 0;
        }
        default: return 0;
    }
}

/* scoreModesMessage, mode_s.c:309-419 */
static int score_message(uint8_t *msg, int validbits) {
    if (validbits < 56) return -2;
    int msgtype = msg[0] >> 3;
    if (validbits >= 112) {
        uint8_t orig = fix_df17(msg, &msgtype);
        if (orig) {
            msg[0] = orig;
            return filter_test(aa_field(msg)) ? 1800 / 2 : 1400 / 2;
        }
    }
    int msgbits = (msgtype & 0x10) ? 112 : 56;
    if (validbits < msgbits) return -2;
    static const uint8_t zeros[7];
    if (!memcmp(zeros, msg, 7)) return -2;
    uint32_t crc = modes_oracle_checksum(msg, msgbits);
    const struct einfo *e;
    uint32_t addr;
    switch (msgtype) {
        case 0: case 4: case 5: case 16: case 20: case 21:
            return filter_test(crc) ? 1000 : -1;
        case 11:
            addr = aa_field(msg);
            if (crc & 0xffff80) {
                e = diagnose(crc, msgbits);
                if (!e) return -2;
                if (e->errors > 1) return -2;
                correct_aa(&addr, e);
                return filter_test(addr) ? 800 : -1;
            }
            if ((crc & 0x7f) == 0) return filter_test(addr) ? 1600 : 750;
            return filter_test(addr) ? 1000 : -1;
        case 17: case 18:
            e = diagnose(crc, msgbits);
            if (!e) return -2;
            addr = aa_field(msg);
            correct_aa(&addr, e);
            return filter_test(addr) ? 1800 / (e->errors + 1) : 1400 / (e->errors + 1);
        default:
            return -2;
    }
}

/* decodeModesMessage CRC/address stage, mode_s.c:443-606 + the filter add at :766-779.
 * msg is corrected in place.  Returns 0 / -1 / -2. */
static int decode_crc_stage(uint8_t *msg, int *msgtype, int *msgbits, int *correctedbits, uint32_t *addr_out) {
    static const uint8_t zeros[7];
    if (!memcmp(zeros, msg, 7)) return -2;
    *msgtype = msg[0] >> 3;
    *correctedbits = 0;
    if (fix_df17(msg, msgtype)) *correctedbits = 1;
    *msgbits = (*msgtype & 0x10) ? 112 : 56;
    uint32_t crc = modes_oracle_checksum(msg, *msgbits);
    uint32_t addr = 0xDEADBEEF;
    int iid = 0;
    const struct einfo *e;
    switch (*msgtype) {
        case 0: case 4: case 5: case 16:
        case 24: case 25: case 26: case 27: case 28: case 29: case 30: case 31:
            if (!filter_test(crc)) return -1;
            addr = crc;
            break;
        case 11:
            iid = crc & 0x7f;
            if (crc & 0xffff80) {
                e = diagnose(crc, *msgbits);
                if (!e) return -2;
                if (e->errors > 1) return -2;
                *correctedbits = e->errors;
                iid = 0;
                checksum_fix(msg, e);
                if (!filter_test(aa_field(msg))) return -1;
            }
            addr = aa_field(msg);
            break;
        case 17: case 18:
            if (crc != 0) {
                e = diagnose(crc, *msgbits);
                if (!e) return -2;
                uint32_t addr1 = aa_field(msg);
                *correctedbits = e->errors;
                checksum_fix(msg, e);
                uint32_t addr2 = aa_field(msg);
                if (addr1 != addr2 && !filter_test(addr2)) return -1;
            }
            addr = aa_field(msg);
            break;
        case 20: case 21:
            if (!filter_test(crc)) return -1;
            addr = crc;
            break;
        default:
            return -2;
    }
    *addr_out = addr;
    /* mode_s.c:766-779: the only place that adds addresses */
    if (!*correctedbits && (*msgtype == 17 || (*msgtype == 11 && iid == 0)))
        filter_add(addr);
    return 0;
}

/* --------------------------------------------------------------- demod_2400.c */

/* slice_phase0..4, demod_2400.c:74-93 */
static const int SLICE_COEF[5][4] = {
    {18, -15, -3, 0}, {14, -5, -9, 0}, {16, 5, -20, 0}, {7, 11, -18, 0}, {4, 15, -20, 1}};

/* Closed form of slice_byte's five switch cases (demod_2400.c:133-213): bit k of a frame tried
 * at phase t (4..8) is the sign of slice_phase[(t + 12k) % 5 ...] — precisely: with
 * u = (t % 5) + 12k, the correlator row is u % 5 and it is applied at sample
 * pa + 19 + t/5 + u/5 (verified against all 5 x 8 table entries). */
static void slice_bytes(const uint16_t *pa, int t, int first, int count, uint8_t *out) {
    for (int by = first; by < first + count; ++by) {
        uint8_t v = 0;
        for (int b = 0; b < 8; ++b) {
            int k = by * 8 + b;
            int u = (t % 5) + 12 * k;
            const uint16_t *s = pa + 19 + t / 5 + u / 5;
            const int *c = SLICE_COEF[u % 5];
            int corr = c[0] * s[0] + c[1] * s[1] + c[2] * s[2] + c[3] * s[3];
            if (corr > 0) v |= 0x80 >> b;
        }
        out[by] = v;
    }
}

static uint32_t g_valid_short, g_valid_long;

/* init_bitsets, demod_2400.c:112-128 (ENABLE_DF24 is off, readsb.h:303) */
static void init_bitsets(void) {
    g_valid_short = (1u << 0) | (1u << 4) | (1u << 5) | (1u << 11);
    g_valid_long = (1u << 16) | (1u << 17) | (1u << 18) | (1u << 20) | (1u << 21);
    if (g_fixDF && g_nfix)
        for (int bit = 0; bit < 5; ++bit) g_valid_long |= 1u << (17 ^ (1 << bit));
    /* generate_damage_set(17,1) also contains 17 itself */
}

struct run {
    struct oracle_msg *out; uint64_t nout, cap;
    struct oracle_stats *st;
    int thr;
    int64_t synthetic_now;
};

/* score_phase, demod_2400.c:215-258 */
static void score_phase(struct run *r, int t, const uint16_t *pa, uint8_t *best, int *bestscore, int *bestphase) {
    r->st->demod_preamblePhase[t - 4]++;
    uint8_t msg[14];
    memset(msg, 0, sizeof(msg));
    slice_bytes(pa, t, 0, 1, msg);
    uint32_t df = msg[0] >> 3;
    int bytelen;
    if (g_valid_long & (1u << df)) bytelen = 14;
    else if (g_valid_short & (1u << df)) bytelen = 7;
    else { if (-2 > *bestscore) *bestscore = -2; return; }
    slice_bytes(pa, t, 1, bytelen - 1, msg);
    int score = score_message(msg, bytelen * 8);
    if (score > *bestscore) {
        memcpy(best, msg, 14);
        *bestscore = score;
        *bestphase = t;
    }
}

/* demodulate2400, demod_2400.c:264-482, on one buffer: m[0..len+TRAILING), block start
 * sampleTimestamp/sysTimestamp as sdr_ifile.c:206,216 computes them. */
static void demod_buffer(struct run *r, const uint16_t *m, uint32_t mlen, int64_t sampleTimestamp,
                         int64_t sysTimestamp, double mean_power) {
    uint64_t sum_scaled_signal_power = 0;
    r->synthetic_now = sysTimestamp;                       /* :283-285 */
    for (uint32_t j = 0; j < mlen; j++) {
        const uint16_t *pa = m + j;
        /* :311-322 — the 10x unrolled pre-check is equivalent to testing every position */
        if (!(pa[1] > pa[7] && pa[12] > pa[14] && pa[12] > pa[15])) continue;
        int32_t base_noise = pa[5] + pa[8] + pa[16] + pa[17] + pa[18];
        int32_t ref_level = (base_noise * r->thr) >> 5;    /* :337-339, no dropped samples for ifile */
        int bestscore = -42, bestphase = 0;
        uint8_t best[14];
        int32_t diff_2_3 = pa[2] - pa[3];
        int32_t sum_1_4 = pa[1] + pa[4];
        int32_t diff_10_11 = pa[10] - pa[11];
        int32_t common3456 = sum_1_4 - diff_2_3 + pa[9] + pa[12];
        if (common3456 - diff_10_11 >= ref_level) {        /* :352-359 */
            score_phase(r, 4, pa, best, &bestscore, &bestphase);
            score_phase(r, 5, pa, best, &bestscore, &bestphase);
        }
        if (common3456 + diff_10_11 >= ref_level) {        /* :364-371 */
            score_phase(r, 6, pa, best, &bestscore, &bestphase);
            score_phase(r, 7, pa, best, &bestscore, &bestphase);
        }
        if (sum_1_4 + 2 * diff_2_3 + diff_10_11 + pa[12] >= ref_level)   /* :376-378 */
            score_phase(r, 8, pa, best, &bestscore, &bestphase);
        if (bestscore == -42) continue;
        r->st->demod_preambles++;
        if (bestscore < 0) {
            if (bestscore == -1) r->st->demod_rejected_unknown_icao++;
            else r->st->demod_rejected_bad++;
            continue;
        }
        int msglen = (best[0] & 0x80) ? 112 : 56;          /* :399, DF before any DF repair */
        int64_t timestamp = sampleTimestamp + (int64_t) j * 5 + (8 + 56) * 12 + bestphase;   /* :406 */
        int64_t msg_sys = sysTimestamp + (timestamp - sampleTimestamp) / 12000;             /* :409 */
        r->synthetic_now = msg_sys;                        /* :412-414 */

        struct oracle_msg o;
        memset(&o, 0, sizeof(o));
        memcpy(o.raw, best, 14);
        memcpy(o.msg, best, 14);
        int result = decode_crc_stage(o.msg, &o.msgtype, &o.msgbits, &o.correctedbits, &o.addr);
        if (result < 0) {                                  /* :423-429: counted, NOT skipped over */
            if (result == -1) r->st->demod_rejected_unknown_icao++;
            else r->st->demod_rejected_bad++;
            continue;
        }
        r->st->demod_accepted[o.correctedbits]++;
        r->st->demod_bestPhase[bestphase - 4]++;
        /* :436-457 */
        uint64_t scaled = 0;
        int signal_len = msglen * 12 / 5;
        for (int k = 0; k < signal_len; ++k) { uint32_t v = pa[19 + k]; scaled += (uint64_t) v * v; }
        double signal_power = scaled / 65535.0 / 65535.0;
        o.signalLevel = signal_power / signal_len;
        r->st->signal_power_sum += signal_power;
        r->st->signal_power_count += signal_len;
        sum_scaled_signal_power += scaled;
        if (o.signalLevel > r->st->peak_signal_power) r->st->peak_signal_power = o.signalLevel;
        if (o.signalLevel > 0.50119) r->st->strong_signal_count++;
        o.timestamp = timestamp;
        o.sys_rel_ms = msg_sys - ORACLE_STARTUP_MS;
        o.score = bestscore;
        if (r->nout == r->cap) { r->cap = r->cap ? r->cap * 2 : 65536; r->out = realloc(r->out, r->cap * sizeof(*r->out)); }
        r->out[r->nout++] = o;
        j += msglen * 8 / 4;                               /* :468, plus the loop's own ++ */
    }
    /* :474-479 */
    double sum_signal_power = sum_scaled_signal_power / 65535.0 / 65535.0;
    r->st->noise_power_sum += (mean_power * mlen - sum_signal_power);
    r->st->noise_power_count += mlen;
}


/* demodulate2400AC, demod_2400.c:575-761, on one buffer (only when enabled with modes_oracle_set_mode_ac:
 * readsb runs it after demodulate2400 on the same buffer, readsb.c:871-874).  Mode A/C bits are 1.45 us
 * apart = 87 cycles of a virtual 60 MHz clock, one 2.4 MHz sample = 25 cycles. */
static int g_mode_ac;
void modes_oracle_set_mode_ac(int on) { g_mode_ac = on; }
/* who flips the ICAO filter: 0 = first flip after buffer 0 (default), 1 = before buffer 0 (the reference program's other
 * start-up order), 2 = only modes_oracle_stream_filter_expire() (same values as mgpu_config.filter_clock) */
static int g_filter_clock;
void modes_oracle_set_filter_clock(int mode) { g_filter_clock = mode; }

static void demod_buffer_ac(struct run *r, const uint16_t *m, uint32_t mlen, int64_t sampleTimestamp,
                            int64_t sysTimestamp, double mean_level, double mean_power) {
    double noise_stddev = sqrt(mean_power - mean_level * mean_level);                 /* :579 */
    unsigned noise_level = (unsigned) ((mean_power + noise_stddev) * 65535 + 0.5);  /* :580 */
    for (unsigned f1_sample = 1; f1_sample < mlen; ++f1_sample) {
        if (!(m[f1_sample - 1] < m[f1_sample + 0])) continue;                        /* :639 rising edge */
        if (m[f1_sample + 2] > m[f1_sample + 0] || m[f1_sample + 2] > m[f1_sample + 1]) continue;   /* :642 */
        unsigned f1_level = (m[f1_sample + 0] + m[f1_sample + 1]) / 2;
        if (noise_level * 2 > f1_level) continue;                                    /* :647, 6 dB above noise */
        float f1a_power = (float) m[f1_sample] * m[f1_sample];                      /* :655-658: float arithmetic */
        float f1b_power = (float) m[f1_sample + 1] * m[f1_sample + 1];
        float fraction = f1b_power / (f1a_power + f1b_power);
        unsigned f1_clock = (unsigned) (25 * (f1_sample + fraction * fraction) + 0.5);
        unsigned f2_clock = f1_clock + (87 * 14);                                    /* :662, F2 is 14 bit periods later */
        unsigned f2_sample = f2_clock / 25;
        if (!(m[f2_sample - 1] < m[f2_sample + 0])) continue;                        /* :666-677: same tests on F2 */
        if (m[f2_sample + 2] > m[f2_sample + 0] || m[f2_sample + 2] > m[f2_sample + 1]) continue;
        unsigned f2_level = (m[f2_sample + 0] + m[f2_sample + 1]) / 2;
        if (noise_level * 2 > f2_level) continue;
        unsigned f1f2_level = (f1_level > f2_level ? f1_level : f2_level);
        float midpoint = sqrtf(noise_level * f1f2_level);                            /* :683 (unsigned product, then float) */
        unsigned signal_threshold = (unsigned) (midpoint * M_SQRT2 + 0.5);           /* +3 dB */
        unsigned noise_threshold = (unsigned) (midpoint / M_SQRT2 + 0.5);            /* -3 dB */
        unsigned uncertain_bits = 0, noisy_bits = 0, bits = 0, bit, clock;
        for (bit = 0, clock = f1_clock; bit < 20; ++bit, clock += 87) {              /* :692-713 */
            unsigned sample = clock / 25;
            bits <<= 1; noisy_bits <<= 1; uncertain_bits <<= 1;
            if (m[sample + 2] >= signal_threshold) noisy_bits |= 1;
            if (m[sample + 0] >= signal_threshold || m[sample + 1] >= signal_threshold) bits |= 1;
            else if (m[sample + 0] > noise_threshold && m[sample + 1] > noise_threshold) uncertain_bits |= 1;
        }
        if ((bits & 0x80020) != 0x80020) continue;                                   /* :716 framing bits on */
        if ((bits & 0x0101B) != 0) continue;                                         /* :721 quiet bits off */
        if (noisy_bits || uncertain_bits) continue;                                  /* :725 */
        unsigned modeac =                                                            /* :731-744 */
                ((bits & 0x40000) ? 0x0010 : 0) | ((bits & 0x20000) ? 0x1000 : 0) | ((bits & 0x10000) ? 0x0020 : 0) |
                ((bits & 0x08000) ? 0x2000 : 0) | ((bits & 0x04000) ? 0x0040 : 0) | ((bits & 0x02000) ? 0x4000 : 0) |
                ((bits & 0x00800) ? 0x0100 : 0) | ((bits & 0x00400) ? 0x0001 : 0) | ((bits & 0x00200) ? 0x0200 : 0) |
                ((bits & 0x00100) ? 0x0002 : 0) | ((bits & 0x00080) ? 0x0400 : 0) | ((bits & 0x00040) ? 0x0004 : 0) |
                ((bits & 0x00004) ? 0x0080 : 0);
        struct oracle_msg o;
        memset(&o, 0, sizeof(o));
        o.timestamp = sampleTimestamp + f2_clock / 5;                                /* :755, 60 MHz -> 12 MHz, at F2 */
        o.sys_rel_ms = sysTimestamp + (o.timestamp - sampleTimestamp) / 12000 - ORACLE_STARTUP_MS;   /* :758 */
        o.msgtype = 77; o.msgbits = 16;                                              /* decodeModeAMessage, mode_ac.c:165-200 */
        o.msg[0] = o.raw[0] = (uint8_t) (modeac >> 8);
        o.msg[1] = o.raw[1] = (uint8_t) modeac;
        o.addr = modeac & 0xFF7F;                                                    /* low 24 bits of (ModeA & 0xFF7F) | MODES_NON_ICAO_ADDRESS */
        if (r->nout == r->cap) { r->cap = r->cap ? r->cap * 2 : 65536; r->out = realloc(r->out, r->cap * sizeof(*r->out)); }
        r->out[r->nout++] = o;
        f1_sample += (20 * 87 / 25);                                                 /* :765 */
        r->st->demod_modeac++;
    }
}

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + ts.tv_nsec * 1e-9;
}

/* ifileRun (sdr_ifile.c:169-270) + decodeEntryPoint (readsb.c:861-902) + backgroundTasks'
 * filter flip (readsb.c:1227-1231), on an in-memory capture. */
int modes_oracle_run(const struct modes_oracle_cfg *cfg, const uint8_t *iq, uint64_t nsamples,
                     struct oracle_msg **out, uint64_t *nout, struct oracle_stats *st,
                     uint16_t *mag_dump) {
    g_nfix = cfg->nfix_crc; g_fixDF = cfg->fixDF;
    modes_oracle_crc_init(cfg->nfix_crc);                  /* readsb.c:306 */
    filter_init();                                         /* readsb.c:307 */
    filter_add(BADDR);                                     /* readsb.c:310 */
    init_bitsets();
    memset(st, 0, sizeof(*st));
    struct run r = {0};
    r.st = st; r.thr = cfg->preamble_threshold; r.synthetic_now = ORACLE_STARTUP_MS;
    const unsigned bps = cfg->format == ORACLE_FMT_UC8 ? 2 : 4;
    uint16_t *bufs[2];
    uint32_t lens[2] = {0, 0};
    for (int i = 0; i < 2; i++) bufs[i] = calloc(BUF_SAMPLES + TRAILING, sizeof(uint16_t));
    if (mag_dump) memset(mag_dump, 0, TRAILING * sizeof(uint16_t));
    int64_t next_flip = 0;
    uint64_t sampleCounter = 0, k = 0;
    int eof = 0;
    if (g_filter_clock == 1) {                             /* first backgroundTasks before buffer 0 (readsb.c:857-902) */
        filter_expire();
        next_flip = r.synthetic_now + FILTER_TTL_MS;
        st->nflips++;
    }
    while (!eof) {
        uint16_t *cur = bufs[k & 1], *last = bufs[(k + 1) & 1];
        uint64_t remain = nsamples - sampleCounter;
        uint32_t slen = remain >= BUF_SAMPLES ? BUF_SAMPLES : (uint32_t) remain;
        if (slen < BUF_SAMPLES) eof = 1;                   /* sdr_ifile.c:223-237 */
        int64_t sampleTimestamp = (int64_t) sampleCounter * 5;              /* :206 */
        if (k > 0 && lens[(k + 1) & 1] >= TRAILING)                          /* :209-213 */
            memcpy(cur, last + lens[(k + 1) & 1], TRAILING * sizeof(uint16_t));
        else
            memset(cur, 0, TRAILING * sizeof(uint16_t));
        int64_t sysTimestamp = sampleTimestamp / 12000 + ORACLE_STARTUP_MS;  /* :216 */
        double mean_level, mean_power;
        double t0 = now_s();
        modes_oracle_convert(cfg->format, iq + sampleCounter * bps, cur + TRAILING, slen, &mean_level, &mean_power);
        double t1 = now_s();
        lens[k & 1] = slen;
        if (mag_dump) memcpy(mag_dump + TRAILING + sampleCounter, cur + TRAILING, slen * sizeof(uint16_t));
        sampleCounter += slen;
        demod_buffer(&r, cur, slen, sampleTimestamp, sysTimestamp, mean_power);
        if (g_mode_ac) demod_buffer_ac(&r, cur, slen, sampleTimestamp, sysTimestamp, mean_level, mean_power);
        double t2 = now_s();
        st->t_convert_s += t1 - t0;
        st->t_demod_s += t2 - t1;
        st->samples_processed += slen;
        st->samples_lost += BUF_SAMPLES - slen;            /* readsb.c:886 */
        if (g_filter_clock != 2 && r.synthetic_now >= next_flip) {   /* readsb.c:1227-1231 */
            filter_expire();
            next_flip = r.synthetic_now + FILTER_TTL_MS;
            st->nflips++;
        }
        k++;
    }
    st->nbuffers = k;
    for (int i = 0; i < 2; i++) free(bufs[i]);
    *out = r.out; *nout = r.nout;
    return 0;
}

/* ---- the same, one struct mag_buf at a time: what the decode thread does per buffer (readsb.c:869-902): demodulate2400(buf),
 * with mode_ac demodulate2400AC(buf), then backgroundTasks' filter flip.  One stream per process (the file's statics). ---- */
static struct run g_stream;
static struct oracle_stats g_stream_stats;
static int64_t g_stream_next_flip;

void modes_oracle_stream_begin(const struct modes_oracle_cfg *cfg, int64_t synthetic_now) {
    g_nfix = cfg->nfix_crc; g_fixDF = cfg->fixDF;
    modes_oracle_crc_init(cfg->nfix_crc);
    filter_init();
    filter_add(BADDR);
    init_bitsets();
    free(g_stream.out);
    memset(&g_stream, 0, sizeof(g_stream));
    memset(&g_stream_stats, 0, sizeof(g_stream_stats));
    g_stream.st = &g_stream_stats;
    g_stream.thr = cfg->preamble_threshold;
    g_stream.synthetic_now = synthetic_now;
    g_stream_next_flip = 0;
    if (g_filter_clock == 1) {
        filter_expire();
        g_stream_next_flip = synthetic_now + FILTER_TTL_MS;
        g_stream_stats.nflips++;
    }
}

/* the host's own icaoFilterExpire() / icaoFilterAdd() forwarded (filter clock 2 = external) */
void modes_oracle_stream_filter_expire(void) { filter_expire(); g_stream_stats.nflips++; }
void modes_oracle_stream_filter_add(uint32_t addr) { filter_add(addr); }

/* data = trailing samples then `length` new ones (struct mag_buf.data); sysTimestamp absolute (oracle_msg.sys_rel_ms comes
 * back relative to ORACLE_STARTUP_MS like everywhere else) */
void modes_oracle_stream_mag_buf(const uint16_t *data, uint32_t length, int64_t sampleTimestamp, int64_t sysTimestamp,
                                 double mean_level, double mean_power) {
    demod_buffer(&g_stream, data, length, sampleTimestamp, sysTimestamp, mean_power);
    if (g_mode_ac) demod_buffer_ac(&g_stream, data, length, sampleTimestamp, sysTimestamp, mean_level, mean_power);
    g_stream_stats.samples_processed += length;
    g_stream_stats.samples_lost += BUF_SAMPLES - length;
    if (g_filter_clock != 2 && g_stream.synthetic_now >= g_stream_next_flip) {   /* readsb.c:1227-1231 */
        filter_expire();
        g_stream_next_flip = g_stream.synthetic_now + FILTER_TTL_MS;
        g_stream_stats.nflips++;
    }
    g_stream_stats.nbuffers++;
}

/* messages accumulated since the last take (caller frees with modes_oracle_free); *st = running counters */
uint64_t modes_oracle_stream_take(struct oracle_msg **out, struct oracle_stats *st) {
    const uint64_t n = g_stream.nout;
    *out = g_stream.out;
    g_stream.out = NULL; g_stream.nout = 0; g_stream.cap = 0;
    if (st) *st = g_stream_stats;
    return n;
}

/* Beast wire format of one accepted message: modesSendBeastOutput, net_io.c:1655-1714 (the frame part; the 0x1a 0xe3
 * receiverId prefix is only sent with --net-receiver-id).  0x1a, type '2' (7 bytes) / '3' (14) / '1' (Mode A/C, 2),
 * the 12 MHz timestamp as 6 bytes big-endian (netTimestamp :1620-1648), one byte of signal level, the message
 * bytes; every 0x1a payload byte doubled.  Returns the number of bytes written (at most 2 + 2 * 21), 0 for a
 * length the format does not carry. */
size_t modes_oracle_beast_frame(const struct oracle_msg *m, uint8_t *out) {
    uint8_t *p = out;
    int msgLen = m->msgbits / 8;
    *p++ = 0x1a;
    if (msgLen == 7) *p++ = '2';
    else if (msgLen == 14) *p++ = '3';
    else if (msgLen == 2) *p++ = '1';
    else return 0;
    for (int sh = 40; sh >= 0; sh -= 8) {                       /* netTimestamp */
        uint8_t ch = (uint8_t) (m->timestamp >> sh);
        *p++ = ch;
        if (ch == 0x1a) *p++ = ch;
    }
    int sig = (int) nearbyint(sqrt(m->signalLevel) * 255);      /* :1696-1700 */
    if (m->signalLevel > 0 && sig < 1) sig = 1;
    if (sig > 255) sig = 255;
    { uint8_t ch = (uint8_t) sig; *p++ = ch; if (ch == 0x1a) *p++ = ch; }
    for (int j = 0; j < msgLen; j++) {                          /* mm->msg: the corrected bytes (no --net-verbatim) */
        uint8_t ch = m->msg[j];
        *p++ = ch;
        if (ch == 0x1a) *p++ = ch;
    }
    return (size_t) (p - out);
}

void modes_oracle_free(void *p) { free(p); }

#ifdef MODES_ORACLE_MAIN
/* modes_oracle_cli <UC8|SC16|SC16Q11> <nfix> <fixdf> <thr> <in.iq> <out.msgs> [out.stats] [out.mag] */
int main(int argc, char **argv) {
    if (argc < 7) {
        fprintf(stderr, "usage: %s <UC8|SC16|SC16Q11> <nfix> <fixdf> <thr> <in.iq> <out.msgs> [out.stats] [out.mag]\n", argv[0]);
        return 2;
    }
    struct modes_oracle_cfg cfg;
    cfg.format = !strcmp(argv[1], "UC8") ? ORACLE_FMT_UC8 : !strcmp(argv[1], "SC16") ? ORACLE_FMT_SC16 : ORACLE_FMT_SC16Q11;
    cfg.nfix_crc = atoi(argv[2]); cfg.fixDF = atoi(argv[3]); cfg.preamble_threshold = atoi(argv[4]);
    FILE *f = fopen(argv[5], "rb");
    if (!f) { perror(argv[5]); return 1; }
    fseek(f, 0, SEEK_END); long sz = ftell(f); fseek(f, 0, SEEK_SET);
    uint8_t *iq = malloc(sz ? sz : 1);
    if (fread(iq, 1, sz, f) != (size_t) sz) { perror("fread"); return 1; }
    fclose(f);
    uint64_t n = sz / (cfg.format == ORACLE_FMT_UC8 ? 2 : 4);
    struct oracle_msg *out; uint64_t nout; struct oracle_stats st;
    uint16_t *mag = argc > 8 ? malloc((n + TRAILING) * sizeof(uint16_t)) : NULL;
    modes_oracle_run(&cfg, iq, n, &out, &nout, &st, mag);
    f = fopen(argv[6], "wb"); fwrite(out, sizeof(*out), nout, f); fclose(f);
    if (argc > 7) { f = fopen(argv[7], "wb"); fwrite(&st, sizeof(st), 1, f); fclose(f); }
    if (argc > 8) { f = fopen(argv[8], "wb"); fwrite(mag, sizeof(uint16_t), n + TRAILING, f); fclose(f); }
    fprintf(stderr, "modes_oracle: %llu samples, %llu msgs, convert %.3f s, demod %.3f s\n",
            (unsigned long long) n, (unsigned long long) nout, st.t_convert_s, st.t_demod_s);
    return 0;
}
#endif
