/* oracle/modes_oracle_gate.c — TEST INFRASTRUCTURE ONLY (see modes_oracle.h): CPU restatement of the first stage of the
 * reference's tracker and of its forwarding rule, as far as both can be decided WITHOUT the position tracker.
 * Pinned by the whole reference program's --dump-beast streams (tests/golden/gate_*.npz, tests/golden/make_gate_golden.py): every
 * verdict this file calls certain equals what the program did; what it cannot know it calls DEFERRED.
 *
 * What the reference does with one accepted message (after decodeModesMessage, in netUseMessage order):
 *   drainMessageBuffer (net_io.c:5924-5940): for a batch of messages — all of one sample buffer's, cut every 256
 *   (netUseMessage drains a full buffer, net_io.c:5996-6005; alloc = 128 * min(3, net_sndbuf_size) = 256, net_io.c:863,
 *   readsb.c:153; demodulate2400 drains at its end, demod_2400.c:481) — FIRST trackUpdateFromMessage() for every message,
 *   THEN outputMessage() for every message.
 *   trackUpdateFromMessage (track.c:1858-2680), the part that decides mm->aircraft and a->messages:
 *     Mode A/C: counted, no aircraft (:1871-1876).   address_reliable = DF17 / DF18 / DF11 with IID 0 (:1688-1693).
 *     a = aircraftGet(addr); unknown: created iff address_reliable, else mm->aircraft stays NULL (:1905-1915).
 *     position-carrying messages (mm->cpr_valid) keep a copy of the aircraft (:1917-1921) ...
 *     address_reliable: a->seen = now (:1928-1930);  now - a->seen > 45 s: mm->aircraft stays NULL (:1933-1936);
 *     a->messages++ (:1966);  ... the whole position tracker ...
 *     ... and put the copy BACK if the position was judged bad or a duplicate (:2625-2627): a->messages and a->seen roll back.
 *     mm->aircraft = a (:2676).
 *   outputMessage (net_io.c:5822-5885): the message leaves (beast / raw / SBS / the --dump-beast file) iff
 *     (mm->crc == 0 && mm->correctedbits == 0) || (mm->aircraft && mm->aircraft->messages > 1) || Mode A/C     (:5846-5849)
 *   with a->messages read AFTER the whole batch's updates.  (The beast and raw outputs also want correctedbits < 2, :5863-5872;
 *   the dump file does not — that test is the caller's, it needs nothing from here.)
 *   removeStaleRange (track.c:2828-2890): an aircraft without a reliable position that was not `seen` for 5 minutes is deleted
 *   by the next periodic run; one WITH a reliable position once that position is an hour old (30 minutes: non-ICAO addresses).
 *
 * What cannot be decided here is whether a position message was rolled back (CPR decoding, speed checks, receiver range:
 * cpr.c + track.c:423-745, out of scope) and whether a silent aircraft had a reliable position.  So every aircraft
 * carries BOUNDS: messages in [lo, hi] and seen in [lo, hi] — lo as if every position message was rolled back and every possible
 * deletion happened, hi as if none — and a verdict is certain when both bounds agree.  In practice only messages inside an
 * aircraft's first two are ever deferred. */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "modes_oracle.h"

#define GATE_BATCH 256                 /* messages per drainMessageBuffer at most */
#define GATE_SEEN_TTL_MS 45000         /* track.c:1933 */
#define GATE_REMOVE_MS (5 * 60 * 1000) /* track.c:2856 */
/* removeStaleRange's other rule (track.c:2835-2866): an aircraft WITH a reliable position is deleted once that position is older than
 * an hour (30 minutes for non-ICAO addresses), however recently it was heard.  Which position was the last reliable one is the
 * position tracker's knowledge; the earliest candidate is the aircraft's first position message: from then + the timeout on, the
 * aircraft may have been deleted and re-created at any moment — the lower bounds restart at every message. */
#define GATE_POS_TIMEOUT_MS (60ll * 60 * 1000)
#define GATE_POS_TIMEOUT_NONICAO_MS (30ll * 60 * 1000)

struct gate_ac {
    uint32_t key;                      /* address + 1 (0 = empty slot); 25 bits: the ME decode may set MODES_NON_ICAO_ADDRESS (1 << 24) */
    uint8_t exists_lo, exists_hi;
    uint32_t cnt_lo, cnt_hi;           /* a->messages, bounds */
    int64_t seen_lo, seen_hi;          /* a->seen, bounds */
    int64_t cpr_first;                 /* the aircraft's first position message (0: none yet) */
};

struct oracle_gate {
    struct gate_ac *tab;
    uint64_t cap, used;
};

struct oracle_gate *modes_oracle_gate_new(void) {
    struct oracle_gate *g = calloc(1, sizeof(*g));
    g->cap = 1u << 16;
    g->tab = calloc(g->cap, sizeof(*g->tab));
    return g;
}

void modes_oracle_gate_free(struct oracle_gate *g) {
    if (g) { free(g->tab); free(g); }
}

static struct gate_ac *gate_slot(struct oracle_gate *g, uint32_t addr, int create) {
    for (;;) {
        uint64_t h = ((uint64_t) (addr + 1) * 0x9E3779B97F4A7C15ull) >> 20 & (g->cap - 1);
        for (;; h = (h + 1) & (g->cap - 1)) {
            if (g->tab[h].key == addr + 1) return &g->tab[h];
            if (g->tab[h].key == 0) break;
        }
        if (!create) return NULL;
        if (g->used * 2 < g->cap) { g->tab[h].key = addr + 1; g->used++; return &g->tab[h]; }
        struct gate_ac *old = g->tab;                        /* grow: re-insert everything */
        const uint64_t oc = g->cap;
        g->cap *= 2; g->used = 0;
        g->tab = calloc(g->cap, sizeof(*g->tab));
        for (uint64_t i = 0; i < oc; ++i)
            if (old[i].key) { struct gate_ac *s = gate_slot(g, old[i].key - 1, 1); *s = old[i]; }
        free(old);
    }
}

/* n messages in netUseMessage order, whole sample buffers per call (buffer[] = index of the buffer a message came out of,
 * non-decreasing; Mode A/C replies of a buffer follow its Mode S messages).
 * verdict[i]: bits 0-1: 0 not forwarded, 1 forwarded, 2 deferred to the tracker; bit 2: address_reliable; bit 3: mm->aircraft
 * may be set; bit 4: mm->aircraft is set for certain */
void modes_oracle_gate_run(struct oracle_gate *g, uint64_t n, const uint8_t *msgtype, const uint32_t *addr, const uint8_t *iid,
                           const uint8_t *correctedbits, const uint8_t *cpr_valid, const int64_t *now_ms, const uint64_t *buffer,
                           uint8_t *verdict) {
    uint64_t i = 0;
    while (i < n) {
        if (msgtype[i] == 77) { verdict[i++] = 1; continue; }
        /* the batch: Mode S messages of this buffer, at most GATE_BATCH (Mode A/C replies in between do not count, they come behind) */
        uint64_t j = i, in_batch = 0;
        while (j < n && (msgtype[j] == 77 || (buffer[j] == buffer[i] && in_batch < GATE_BATCH))) { if (msgtype[j] != 77) ++in_batch; ++j; }
        for (uint64_t k = i; k < j; ++k) {                       /* trackUpdateFromMessage for all of them */
            if (msgtype[k] == 77) { verdict[k] = 1; continue; }
            const int reliable = msgtype[k] == 17 || msgtype[k] == 18 || (msgtype[k] == 11 && iid[k] == 0);
            uint8_t v = reliable ? 4 : 0;
            struct gate_ac *s = gate_slot(g, addr[k], reliable);
            if (s && s->exists_lo && now_ms[k] - s->seen_lo > GATE_REMOVE_MS) {   /* may have been deleted meanwhile */
                s->exists_lo = 0; s->cnt_lo = 0; s->seen_lo = 0;
            }
            if (s && s->cpr_first && now_ms[k] - s->cpr_first > ((addr[k] & (1u << 24)) ? GATE_POS_TIMEOUT_NONICAO_MS : GATE_POS_TIMEOUT_MS)) {
                s->exists_lo = 0; s->cnt_lo = 0; s->seen_lo = 0;     /* ... or its reliable position may have timed out */
            }
            if (reliable) {
                s->exists_lo = s->exists_hi = 1;                 /* (aircraftCreate is not rolled back) */
                s->seen_hi = now_ms[k]; s->cnt_hi++;
                if (!cpr_valid[k]) { s->seen_lo = now_ms[k]; s->cnt_lo++; }
                else if (!s->cpr_first) s->cpr_first = now_ms[k];
                v |= 8 | 16;
            } else if (s && s->exists_hi) {
                const int ok_lo = s->exists_lo && now_ms[k] - s->seen_lo <= GATE_SEEN_TTL_MS;
                const int ok_hi = now_ms[k] - s->seen_hi <= GATE_SEEN_TTL_MS;
                if (ok_lo) s->cnt_lo++;
                if (ok_lo || ok_hi) { s->cnt_hi++; v |= 8; }
                if (ok_lo) v |= 16;
            }
            verdict[k] = v;
        }
        for (uint64_t k = i; k < j; ++k) {                       /* outputMessage for all of them */
            if (msgtype[k] == 77) continue;
            const int reliable = (verdict[k] & 4) != 0;
            const int crc_zero = reliable || (msgtype[k] != 11 && msgtype[k] != 17 && msgtype[k] != 18 && (addr[k] & 0xffffffu) == 0);   /* Address/Parity formats: crc = address */
            uint8_t out;
            if (crc_zero && correctedbits[k] == 0) out = 1;
            else if (!(verdict[k] & 8)) out = 0;
            else {
                const struct gate_ac *s = gate_slot(g, addr[k], 0);
                if ((verdict[k] & 16) && s->cnt_lo > 1) out = 1;
                else if (s->cnt_hi <= 1) out = 0;
                else out = 2;
            }
            verdict[k] = (uint8_t) ((verdict[k] & ~3u) | out);
        }
        i = j;
    }
}
