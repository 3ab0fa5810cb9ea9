// resolve.cpp — ordered accept / skip-ahead / ICAO-filter walk (see resolve.h).
#include "resolve.h"

#include <algorithm>
#include <cstring>

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

static inline void flip_bit(uint8_t *msg, int bit) { msg[bit >> 3] ^= (uint8_t) (0x80u >> (bit & 7)); }

int64_t Resolver::walk(const PhaseRec *recs, const unsigned long long *sig, uint64_t nrecs,
                       const std::vector<BufferClock> &buffers, std::vector<mgpu_msg> &out, uint32_t *aux_pos,
                       uint16_t *aux_skip, uint32_t *aux_limit, uint64_t aux_cap, ResolveCounts &c) {
    uint64_t i = 0, nout = 0;
    for (const BufferClock &b : buffers) {
        synthetic_now_ = b.sysTimestamp;                       // demod_2400.c:283-285
        const uint64_t end = (uint64_t) b.first + b.length;
        int64_t skip_until = -1;                               // the skip never crosses a buffer (loop-local pa)
        while (i < nrecs && recs[i].pos < end) {
            const uint32_t pos = recs[i].pos;
            uint64_t j = i;
            bool has_uncond = false;
            while (j < nrecs && recs[j].pos == pos) { has_uncond |= !(recs[j].flags & REC_COND); ++j; }
            if ((int64_t) pos <= skip_until) {
                if (has_uncond) ++c.skipped_uncond_groups; else ++c.skipped_cond_groups;
                i = j;
                continue;
            }
            ++c.visited_groups;
            if (has_uncond) ++c.visited_uncond_groups; else ++c.visited_cond_groups;
            // best over the tried phases, in phase order, strict '>' (demod_2400.c:246)
            int best = -2;
            const PhaseRec *br = nullptr;
            bool best_known = false;
            for (uint64_t k = i; k < j; ++k) {
                const PhaseRec &r = recs[k];
                bool known = false;
                int s;
                if (r.score_known == r.score_unknown) s = r.score_known;
                else { known = filter_.test(r.addr); s = known ? r.score_known : r.score_unknown; }
                if (s > best) { best = s; br = &r; best_known = known || r.score_known == r.score_unknown; }
            }
            i = j;
            if (best < 0) {                                    // demod_2400.c:390-397
                if (best == -1) ++c.rejected_unknown; else ++c.rejected_bad;
                continue;
            }
            const int msglen = (br->msg[0] & 0x80) ? 112 : 56; // :399, DF as sliced
            mgpu_msg m;
            std::memset(&m, 0, sizeof(m));
            m.timestamp = b.sampleTimestamp + (int64_t) (pos - b.first) * 5 + (8 + 56) * 12 + br->phase;   // :406
            m.sysTimestamp = b.sysTimestamp + (m.timestamp - b.sampleTimestamp) / 12000;                    // :409
            synthetic_now_ = m.sysTimestamp;                   // :412-414
            // decodeModesMessage's CRC/address stage: same filter state as the scoring above
            bool accept = (br->flags & REC_ACCEPT_IF_UNKNOWN) != 0;
            if (!accept) accept = (br->score_known == br->score_unknown) ? best_known : filter_.test(br->addr);
            if (!accept) { ++c.rejected_unknown; continue; }   // :423-429, no skip-ahead
            m.score = (int16_t) best;
            m.phase = br->phase;
            m.correctedbits = (br->flags >> REC_CORR_SHIFT) & 3;
            const int raw_bytes = (br->flags & REC_LONG) ? 14 : 7;
            std::memcpy(m.raw, br->msg, raw_bytes);
            std::memcpy(m.msg, br->msg, raw_bytes);
            if (br->flags & REC_DFFIX) m.msg[0] = (uint8_t) ((m.msg[0] & 7) | (17 << 3));
            else {
                if (br->fixbit0 != 0xff) flip_bit(m.msg, br->fixbit0);
                if (br->fixbit1 != 0xff) flip_bit(m.msg, br->fixbit1);
            }
            m.msgtype = m.msg[0] >> 3;
            m.msgbits = (m.msgtype & 0x10) ? 112 : 56;
            if (m.msgbits == 56) { std::memset(m.msg + 7, 0, 7); std::memset(m.raw + 7, 0, 7); }
            m.addr = br->addr & 0xffffffu;
            m.sig_len = (uint16_t) (msglen * 12 / 5);          // :439
            m.sig_sumsq = sig[br - recs];                      // :442-445, precomputed per record on the GPU
            if (br->flags & REC_ADDER) filter_.add(m.addr);    // mode_s.c:766-779
            ++c.accepted[m.correctedbits];
            ++c.best_phase[br->phase - 4];
            if (nout >= aux_cap) return -1;
            out.push_back(m);
            aux_pos[nout] = pos;
            aux_skip[nout] = (uint16_t) (msglen * 8 / 4);     // :468
            aux_limit[nout] = (uint32_t) end;
            ++nout;
            skip_until = (int64_t) pos + msglen * 8 / 4;
        }
        after_buffer();
    }
    return (int64_t) nout;
}

}  // namespace mgpu
