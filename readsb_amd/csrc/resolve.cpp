// resolve.cpp — ordered accept / skip-ahead / ICAO-filter walk (see resolve.h).
#include "resolve.h"

#include <algorithm>
#include <cstddef>
#include <cstring>
#if defined(__SSE2__)
#include <emmintrin.h>
#endif

namespace mgpu {

static constexpr uint32_t kShowOnlyDefault = 0xff123456u;   // BADDR, readsb.h:296; modesInit adds it (readsb.c:310)
static constexpr int64_t kFilterTtlMs = 60000;              // MODES_ICAO_FILTER_TTL, readsb.h:315

IcaoFilter::IcaoFilter() {
    for (int g = 0; g < 2; ++g) bits_[g].assign(1u << 18, 0);
    init();
}

void IcaoFilter::clear_gen(int g) {
    for (uint32_t a : members_[g]) bits_[g][a >> 6] = 0;
    members_[g].clear();
    big_[g].clear();
}

void IcaoFilter::init() {
    clear_gen(0);
    clear_gen(1);
    active_ = 0;
    occupied_ = 0;
    filter_bits_ = 8;
}

bool IcaoFilter::big_test(uint32_t addr) const {
    for (int g = 0; g < 2; ++g)
        if (std::find(big_[g].begin(), big_[g].end(), addr) != big_[g].end()) return true;
    return false;
}

void IcaoFilter::resize(uint32_t bits) {
    // two fresh tables; only the ACTIVE generation's entries are re-inserted
    filter_bits_ = bits;
    clear_gen(active_ ^ 1);
    occupied_ = (uint32_t) (members_[active_].size() + big_[active_].size());
}

void IcaoFilter::add(uint32_t addr) {
    bool inserted = false;
    if (addr < (1u << 24)) {
        uint64_t &w = bits_[active_][addr >> 6];
        const uint64_t bit = 1ull << (addr & 63);
        if (!(w & bit)) { w |= bit; members_[active_].push_back(addr); inserted = true; }
    } else if (std::find(big_[active_].begin(), big_[active_].end(), addr) == big_[active_].end()) {
        big_[active_].push_back(addr);
        inserted = true;
    }
    if (inserted) ++occupied_;
    if (occupied_ > (1u << filter_bits_) / 3 && filter_bits_ < 20) resize(filter_bits_ + 1);
}

void IcaoFilter::expire() {
    if (occupied_ < (1u << filter_bits_) / 9 && filter_bits_ > 8) resize(filter_bits_ - 1);
    occupied_ = 0;
    clear_gen(active_ ^ 1);
    active_ ^= 1;
}

void Resolver::reset(int64_t startup_ms) {
    filter_.init();
    filter_.add(kShowOnlyDefault);
    synthetic_now_ = startup_ms;   // Modes.synthetic_now armed by ifileOpen (sdr_ifile.c:131-133)
    next_flip_ = 0;                // static next_flip = 0 (readsb.c:1227)
    nflips_ = 0;
}

void Resolver::after_buffer() {
    // backgroundTasks(now = mstime()) after every buffer (readsb.c:899-902, 1227-1231)
    if (synthetic_now_ >= next_flip_) {
        filter_.expire();
        next_flip_ = synthetic_now_ + kFilterTtlMs;
        ++nflips_;
    }
}

void Resolver::tick_empty(int64_t sysTimestamp) {
    synthetic_now_ = sysTimestamp;   // demod_2400.c:283-285
    after_buffer();
}

static inline int frame_bits(const PhaseRec &r) { return (r.msg[0] & 0x80) ? 112 : 56; }   // demod_2400.c:399, DF as sliced

int64_t Resolver::decide(const PhaseRec *recs, uint64_t nrecs, const std::vector<BufferClock> &buffers, std::vector<Accepted> &acc,
                         uint32_t *aux_pos, uint16_t *aux_skip, uint32_t *aux_limit, uint64_t aux_cap, ResolveCounts &counts) {
    // recs[nrecs] is a sentinel the caller provides (pos = 0xFFFFFFFF)
    (void) nrecs;
    ResolveCounts c;                                           // locals: the optimiser keeps them out of memory
    if (acc.size() < aux_cap) acc.resize(aux_cap);             // a pre-sized array, its length is the return value
    Accepted *out = acc.data();
    const PhaseRec *r = recs;
    uint64_t nout = 0;
    for (uint32_t bi = 0; bi < (uint32_t) buffers.size(); ++bi) {
        const BufferClock &b = buffers[bi];
        synthetic_now_ = b.sysTimestamp;                       // demod_2400.c:283-285
        const uint64_t end = (uint64_t) b.first + b.length;
        int64_t skip_until = -1;                               // the skip never crosses a buffer (loop-local pa)
        while (r->pos < end) {
            const uint32_t pos = r->pos;
            if ((int64_t) pos <= skip_until) {                 // hidden by the previous frame (:468)
                uint32_t all_cond = REC_COND;
                do { all_cond &= r->flags; ++r; } while (r->pos == pos);
                if (all_cond) ++c.skipped_cond_groups; else ++c.skipped_uncond_groups;
                continue;
            }
            // best over the tried phases, in phase order, strict '>' (demod_2400.c:246)
            int best = -2;
            const PhaseRec *br = nullptr;
            bool best_known = false;
            uint32_t all_cond = REC_COND;
            do {
                all_cond &= r->flags;
                const bool fixed = r->score_known == r->score_unknown;
                const bool known = fixed || filter_.test(r->addr);     // for fixed scores either answer scores the same
                const int s = known ? r->score_known : r->score_unknown;
                if (s > best) { best = s; br = r; best_known = known; }
                ++r;
            } while (r->pos == pos);
            ++c.visited_groups;
            if (all_cond) ++c.visited_cond_groups; else ++c.visited_uncond_groups;
            if (best < 0) {                                    // demod_2400.c:390-397
                if (best == -1) ++c.rejected_unknown; else ++c.rejected_bad;
                continue;
            }
            const int msglen = frame_bits(*br);
            const int64_t rel = (int64_t) (pos - b.first) * 5 + (8 + 56) * 12 + br->phase;   // timestamp - sampleTimestamp, :406
            synthetic_now_ = b.sysTimestamp + rel / 12000;                                   // :409-414
            // decodeModesMessage's CRC/address stage: same filter state as the scoring above
            bool accept = (br->flags & REC_ACCEPT_IF_UNKNOWN) != 0;
            if (!accept) accept = best_known;
            if (!accept) { ++c.rejected_unknown; continue; }   // :423-429, no skip-ahead
            if (br->flags & REC_ADDER) filter_.add(br->addr & 0xffffffu);   // mode_s.c:766-779
            ++c.accepted[(br->flags >> REC_CORR_SHIFT) & 3];
            ++c.best_phase[br->phase - 4];
            if (nout >= aux_cap) { counts.add(c); return -1; }
            out[nout] = Accepted{(uint32_t) (br - recs), bi, best};
            aux_pos[nout] = pos;
            aux_skip[nout] = (uint16_t) (msglen * 8 / 4);     // :468
            aux_limit[nout] = (uint32_t) end;
            ++nout;
            skip_until = (int64_t) pos + msglen * 8 / 4;
        }
        after_buffer();
    }
    counts.add(c);
    return (int64_t) nout;
}

void Resolver::build_messages(const PhaseRec *recs, const unsigned long long *sig, const std::vector<BufferClock> &buffers,
                              const Accepted *acc, uint64_t nacc, mgpu_msg *out) {
    static_assert(sizeof(PhaseRec) == 32 && offsetof(PhaseRec, msg) == 16, "frame bytes are the record's second half");
    for (uint64_t n = 0; n < nacc; ++n) {
        const PhaseRec &r = recs[acc[n].rec];
        const BufferClock &b = buffers[acc[n].buffer];
        // the frame as two little-endian words: byte k of the frame = bits 8k.. of lo (k < 8) / hi (k >= 8)
        uint64_t lo, hi;
        std::memcpy(&lo, r.msg, 8);
        std::memcpy(&hi, r.msg + 8, 8);
        if (r.flags & REC_LONG) hi &= 0x0000ffffffffffffull; else { lo &= 0x00ffffffffffffffull; hi = 0; }
        uint64_t mlo = lo, mhi = hi;                           // msg = raw with the CRC repair applied
        if (r.flags & REC_DFFIX) mlo = (mlo & ~0xf8ull) | (17u << 3);      // fixDF17msgtype, mode_s.c:276-301
        else {
            if (r.fixbit0 != 0xff) { const uint64_t f = 0x80ull >> (r.fixbit0 & 7); if (r.fixbit0 < 64) mlo ^= f << (r.fixbit0 & 56); else mhi ^= f << (r.fixbit0 & 56); }
            if (r.fixbit1 != 0xff) { const uint64_t f = 0x80ull >> (r.fixbit1 & 7); if (r.fixbit1 < 64) mlo ^= f << (r.fixbit1 & 56); else mhi ^= f << (r.fixbit1 & 56); }
        }
        const unsigned msgtype = (unsigned) (mlo & 0xff) >> 3;
        const unsigned msgbits = (msgtype & 0x10) ? 112 : 56;
        if (msgbits == 56) { mlo &= 0x00ffffffffffffffull; mhi = 0; lo &= 0x00ffffffffffffffull; hi = 0; }
        mgpu_msg m;
        m.timestamp = b.sampleTimestamp + (int64_t) (r.pos - b.first) * 5 + (8 + 56) * 12 + r.phase;   // demod_2400.c:406
        m.sysTimestamp = b.sysTimestamp + (m.timestamp - b.sampleTimestamp) / 12000;                    // :409
        m.sig_sumsq = sig[acc[n].rec];                         // :442-445, precomputed per record on the GPU
        m.sig_len = (uint16_t) (frame_bits(r) * 12 / 5);       // :439
        m.score = (int16_t) acc[n].score;
        m.phase = r.phase;
        m.correctedbits = (r.flags >> REC_CORR_SHIFT) & 3;
        m.msgtype = (uint8_t) msgtype;
        m.msgbits = (uint8_t) msgbits;
        m.addr = r.addr & 0xffffffu;
        std::memcpy(m.msg, &mlo, 8); std::memcpy(m.msg + 8, &mhi, 6);
        std::memcpy(m.raw, &lo, 8); std::memcpy(m.raw + 8, &hi, 6);
#if defined(__SSE2__)
        if ((reinterpret_cast<uintptr_t>(out) & 15) == 0) {   // streaming stores: the consumer is another core, skip the RFO
            __m128i v[4];
            std::memcpy(v, &m, 64);
            __m128i *d = reinterpret_cast<__m128i *>(out + n);
            _mm_stream_si128(d + 0, v[0]); _mm_stream_si128(d + 1, v[1]);
            _mm_stream_si128(d + 2, v[2]); _mm_stream_si128(d + 3, v[3]);
        } else
#endif
        out[n] = m;
    }
#if defined(__SSE2__)
    _mm_sfence();
#endif
}

int64_t Resolver::walk(const PhaseRec *recs, const unsigned long long *sig, uint64_t nrecs,
                       const std::vector<BufferClock> &buffers, std::vector<mgpu_msg> &out, uint32_t *aux_pos,
                       uint16_t *aux_skip, uint32_t *aux_limit, uint64_t aux_cap, ResolveCounts &c) {
    std::vector<Accepted> acc;
    const int64_t n = decide(recs, nrecs, buffers, acc, aux_pos, aux_skip, aux_limit, aux_cap, c);
    if (n <= 0) return n;
    const size_t first = out.size();
    out.resize(first + (size_t) n);
    build_messages(recs, sig, buffers, acc.data(), (uint64_t) n, out.data() + first);
    return n;
}

}  // namespace mgpu
