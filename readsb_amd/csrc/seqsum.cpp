// seqsum.cpp — the reference's SEQUENTIAL double sums, re-added exactly by ranges (config 5 with the walk sharded over ranks).
//
// demodulate2400 adds every accepted message's signal power into one double, message after message (demod_2400.c:445-447:
// `sum_signal_power += signal_power` per message, folded into Modes.stats_current.signal_power_sum per buffer); every addition
// rounds, so the total depends on the order and cannot be had by adding the ranges' partial sums.  Re-adding 8 M terms on the rank
// that combines the ranges is a dependent chain of 8 M additions (6-9 ms for the one-hour capture: more than that rank's whole
// share of the walk).  What makes the chain parallel is what made k_fsum_sc16's (kernels/convert.inc): while the running sum s
// stays inside one binade [2^e, 2^(e+1)) every addition rounds to the same grid g = 2^(e-52); with s = S g and x = (d + f) g,
// RN(s + x) = (S + d + [f > 1/2]) g unless f = 1/2 exactly (a tie: round half to even, i.e. the carry is the parity of S + d).
// d + [f > 1/2] is what RN(2^e + x) - 2^e holds, and the residual x - (RN(2^e + x) - 2^e) tells a tie (|r| = g / 2).  A tie
// always leaves the sum even, so every tie's carry but a block's first depends only on the steps since the tie before it.
//     A BLOCK of terms is therefore one integer addition — total steps, plus (first tie only) one bit of the incoming sum —
//     valid when the incoming sum is in the binade the block was prepared for and the block does not leave it.
// The ranks prepare their blocks side by side against the binades an APPROXIMATE running sum predicts (their own sequential sum
// on top of the earlier ranges' plain totals); the combining rank applies ~8000 blocks instead of 8 M additions and re-adds, term
// by term, only the blocks whose premise fails (the ~20 in which the sum crosses into the next binade, the stream's first).
#include <cmath>
#include <cstdint>
#include <functional>
#include <thread>
#include <vector>

#include "../../include/modes_gpu.h"

namespace {
inline bool has_power(const mgpu_msg &m) { return m.msgtype != 77; }                 // (Mode A/C replies carry no signal power)
inline double power_of(const mgpu_msg &m) { return (double) m.sig_sumsq / 65535.0 / 65535.0; }   // demod_2400.c:445
constexpr uint32_t kValid = 1, kTie = 2, kParity = 4;
}

// the terms of a range: its messages, or (round 6) the 8-byte numerators of their signal powers the builder logged (mgpu_shard_signal_terms)
struct MsgTerms {
    const struct mgpu_msg *m;
    bool has(uint64_t i) const { return has_power(m[i]); }
    double at(uint64_t i) const { return power_of(m[i]); }
};
struct SumsqTerms {
    const uint64_t *q;
    bool has(uint64_t) const { return true; }
    double at(uint64_t i) const { return (double) q[i] / 65535.0 / 65535.0; }
};

// One part of a range: its blocks against the binades `pred` (the approximate running sum where the part begins) predicts.
template <class Terms>
static void prepare_part(double pred, const Terms msgs, uint64_t lo0, uint64_t hi0, uint32_t block, struct mgpu_sum_block *out) {
    for (uint64_t lo = lo0, b = lo0 / block; lo < hi0; lo += block, ++b) {
        const uint64_t hi = lo + block < hi0 ? lo + block : hi0;
        mgpu_sum_block sb{0, 0, 0};
        if (!(pred > 0.0) || !std::isfinite(pred)) {            // no binade yet (the stream's first block): re-added term by term
            for (uint64_t i = lo; i < hi; ++i) if (msgs.has(i)) pred += msgs.at(i);
            out[b] = sb;
            continue;
        }
        const int e = std::ilogb(pred);
        const double M = std::ldexp(1.0, e), halfg = std::ldexp(1.0, e - 53), to_steps = std::ldexp(1.0, 52 - e);
        uint64_t total = 0;
        uint32_t par = 0, flags = kValid;
        for (uint64_t i = lo; i < hi; ++i) {
            if (!msgs.has(i)) continue;
            const double x = msgs.at(i);
            pred += x;
            if (!(x < M)) { flags &= ~kValid; continue; }       // a term as large as the binade itself: the sum leaves it for sure
            const double t = M + x;                             // rounds x to the binade's grid (a tie: to the even side of 2^e + x)
            const double u = t - M;                             // exact
            const double r = x - u;                             // exact, |r| <= g / 2
            const bool tie = std::fabs(r) == halfg;
            const uint64_t q = (uint64_t) (u * to_steps) - ((tie && r < 0.0) ? 1u : 0u);   // a tie's steps without its carry
            total += q;
            par ^= (uint32_t) (q & 1u);
            if (tie) {
                if (!(flags & kTie)) flags |= kTie | (par ? kParity : 0u);   // the block's first tie: its carry needs the incoming sum's parity
                else total += par;                               // every later one: the parity of the steps since the tie before it
                par = 0;                                        // a tie leaves the sum even
            }
        }
        sb.total = total; sb.e = e; sb.flags = flags;
        out[b] = sb;
    }
}

template <class Terms>
static int seqsum_blocks(double approx_start, const Terms msgs, uint64_t n, uint32_t block, struct mgpu_sum_block *out) {
    // the prediction only has to be roughly right, so a range splits into parts prepared side by side: plain part sums first (any
    // order), then every part against the prefix of those
    const uint64_t nblocks = (n + block - 1) / block;
    unsigned parts = std::thread::hardware_concurrency() >= 16 ? 8u : 2u;
    if (nblocks < 64) parts = 1;
    std::vector<double> sums(parts, 0.0);
    std::vector<uint64_t> cut(parts + 1);
    for (unsigned p = 0; p <= parts; ++p) { const uint64_t c = nblocks * p / parts * block; cut[p] = c < n ? c : n; }
    cut[parts] = n;
    auto run = [&](const std::function<void(unsigned)> &f) {
        std::vector<std::thread> th;
        for (unsigned p = 1; p < parts; ++p) th.emplace_back(f, p);
        f(0);
        for (auto &t : th) t.join();
    };
    if (parts > 1)
        run([&](unsigned p) { double s = 0; for (uint64_t i = cut[p]; i < cut[p + 1]; ++i) if (msgs.has(i)) s += msgs.at(i); sums[p] = s; });
    std::vector<double> start(parts, approx_start);
    for (unsigned p = 1; p < parts; ++p) start[p] = start[p - 1] + sums[p - 1];
    run([&](unsigned p) { prepare_part(start[p], msgs, cut[p], cut[p + 1], block, out); });
    return MGPU_OK;
}

extern "C" {

double mgpu_seqsum(double start, const double *terms, uint64_t n) {
    double s = start;
    for (uint64_t i = 0; i < n; ++i) s += terms[i];
    return s;
}

double mgpu_seqsum_signal_power(double start, const struct mgpu_msg *msgs, uint64_t n) {
    double s = start;
    for (uint64_t i = 0; i < n; ++i)
        if (has_power(msgs[i])) s += power_of(msgs[i]);
    return s;
}

int mgpu_seqsum_blocks(double approx_start, const struct mgpu_msg *msgs, uint64_t n, uint32_t block, struct mgpu_sum_block *out) {
    if (!block || (n && (!msgs || !out))) return MGPU_E_INVAL;
    return seqsum_blocks(approx_start, MsgTerms{msgs}, n, block, out);
}

int mgpu_seqsum_blocks_terms(double approx_start, const uint64_t *sumsq, uint64_t n, uint32_t block, struct mgpu_sum_block *out) {
    if (!block || (n && (!sumsq || !out))) return MGPU_E_INVAL;
    return seqsum_blocks(approx_start, SumsqTerms{sumsq}, n, block, out);
}

double mgpu_seqsum_apply(double start, const struct mgpu_msg *msgs, uint64_t n, uint32_t block, const struct mgpu_sum_block *blocks,
                         uint64_t *fallbacks) {
    double s = start;
    uint64_t fb = 0;
    if (!block) return s;
    for (uint64_t lo = 0, b = 0; lo < n; lo += block, ++b) {
        const uint64_t hi = lo + block < n ? lo + block : n;
        const mgpu_sum_block &sb = blocks[b];
        if ((sb.flags & kValid) && s > 0.0 && std::isfinite(s) && std::ilogb(s) == sb.e) {
            const uint64_t S = (uint64_t) std::ldexp(s, 52 - sb.e);           // exact: s is a multiple of the grid
            const uint64_t T = sb.total + ((sb.flags & kTie) ? ((S + ((sb.flags & kParity) ? 1u : 0u)) & 1u) : 0u);
            if (S + T < (1ull << 53)) { s = std::ldexp((double) (S + T), sb.e - 52); continue; }
        }
        ++fb;                                                   // the premise fails: this block term by term
        for (uint64_t i = lo; i < hi; ++i) if (has_power(msgs[i])) s += power_of(msgs[i]);
    }
    if (fallbacks) *fallbacks = fb;
    return s;
}

}  // extern "C"
