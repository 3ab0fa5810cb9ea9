"""CPU: the arithmetic behind k_fsum_sc16 (readsb_amd/csrc/kernels/convert.inc) restated in numpy and checked against what it
replaces — a sequential float32 running sum, `sum += x` per sample as convert_sc16_nodc / convert_sc16q11_nodc do
(convert.c:225-249, 342-366).  No GPU: this pins the algorithm (grid steps from RN(2^e + x), ties resolved by step parity since
the previous tie, one bit of the incoming sum for a block's first tie, in-order addition where the sum leaves its binade); the
kernel itself is compared with the reference's converters in tests/test_gpu_convert.py."""
import numpy as np
import pytest

F = np.float32


def sequential(xs, s0=F(0)):
    s = F(s0)
    for x in xs:
        s = F(s + x)
    return s


def block_steps(xs, e_bits):
    """Phase A for one block: (total with the first tie's carry taken for an even incoming sum, has_tie, Z at the first tie)."""
    M = np.uint32(e_bits).view(F)
    halfg = np.uint32(e_bits - (24 << 23)).view(F)
    t = (M + xs).astype(F)
    d = t.view(np.uint32).astype(np.int64) - int(e_bits)
    r = (xs - (t - M).astype(F)).astype(F)
    tie = np.abs(r) == halfg
    d = d - (tie & (r < 0))
    if (d >= (1 << 22)).any():
        return None                                   # a sample that large: the block leaves the binade, added in order
    z = np.bitwise_xor.accumulate(d & 1)              # parity of the steps from the block's start through each sample
    total = int(d.sum())
    idx = np.flatnonzero(tie)
    zp = 0                                            # the block's first tie: as if the incoming sum were even
    for i in idx:
        total += int(z[i]) ^ zp
        zp = int(z[i])
    return total, idx.size > 0, (int(z[idx[0]]) if idx.size else 0)


def model_sum(xs, block=512, s0=F(0)):
    s = F(s0)
    for lo in range(0, len(xs), block):
        xb = xs[lo:lo + block]
        sbits = int(np.float32(s).view(np.uint32))
        e_bits = sbits & 0x7f800000
        res = block_steps(xb, e_bits) if e_bits else None
        if res is not None:
            total, has_tie, z_first = res
            S = (sbits & 0x7fffff) | 0x800000
            if has_tie and (S & 1):
                total += -1 if z_first else 1         # an odd incoming sum turns the first tie's carry round
            S2 = S + total
            if S2 < (1 << 24):
                s = np.uint32(e_bits | (S2 & 0x7fffff)).view(F)
                continue
        s = sequential(xb, s)                         # the sum is still zero, or the block leaves the binade
    return s


def _cases():
    rng = np.random.default_rng(1234)
    n = 131072 // 4
    # SC16Q11-like squared magnitudes: small integers x 2^-22 (ties everywhere once the grid is a few bits coarser)
    for scale in (30, 300, 3000, 40000):
        sq = rng.integers(0, scale, size=n).astype(np.int64)
        yield f"magsq small integers < {scale}", (sq.astype(np.float64) * 2.0 ** -22).astype(F)
    # their square roots: full mantissas, ties rare
    sq = rng.integers(0, 5000, size=n).astype(np.float64) * 2.0 ** -22
    yield "mag of small integers", np.sqrt(sq).astype(F)
    # SC16: (I^2 + Q^2) / 2^30 rounded through floats
    i, q = rng.integers(-3000, 3000, size=(2, n)).astype(F) / F(32768.0)
    yield "sc16 magsq", np.minimum(i * i + q * q, F(1)).astype(F)
    # quiet start, zeros, a clipped burst, then noise: binade crossings early and late
    x = (rng.integers(0, 200, size=n).astype(np.float64) * 2.0 ** -22).astype(F)
    x[:3000] = 0
    x[9000:9050] = 1.0
    yield "zeros, burst, noise", x
    #  ONE isolated strong sample while the running sum is still small: its grid steps are 2^22 and more, the block
    # must be added in order (the kernel once clamped the count and took the one-addition path: 0.0511 for 1.0355)
    for k, (quiet, strong) in enumerate(((0.033, (1.0,)), (0.033, (0.02,)), (0.3, (1.0, 1.0)), (0.07, (0.5, 1.0, 0.25)))):
        x = np.full(4096, F(quiet / 2048), dtype=F)
        for j, v in enumerate(strong):
            x[2048 + 600 + 37 * j] = F(v)
        yield f"isolated strong samples {k}", x
    # exact ties by construction against a sum that sits in [4, 8): every sample is half a grid step
    yield "all ties", np.concatenate([np.full(4, F(1.0)), np.full(5000, F(2.0 ** -22))]).astype(F)


@pytest.mark.parametrize("name,xs", list(_cases()), ids=[c[0] for c in _cases()])
def test_block_sum_equals_sequential_float_sum(name, xs):
    want = sequential(xs)
    for block in (256, 512):
        got = model_sum(xs, block)
        assert got.view(np.uint32) == want.view(np.uint32), (name, block, float(got), float(want))


def model_sum_wide(xs, block=512, rel_err=1e-4, seed=0):
    """Round 4's wide form (k_fsum_approx / k_fsum_prep / k_fsum_apply): every block is summarised against the binade an APPROXIMATE
    prefix of block sums predicts — independently of the others —, and the chain only applies the summaries; a block whose
    premise fails (the sum is not in the predicted binade, the block leaves it, an oversized sample) is added exactly."""
    rng = np.random.default_rng(seed)
    nblk = (len(xs) + block - 1) // block
    approx = np.array([float(np.sum(xs[b * block:(b + 1) * block].astype(np.float64))) * (1 + rel_err * rng.uniform(-1, 1)) for b in range(nblk)], dtype=F)
    summaries = []
    for b in range(nblk):                                  # (on the device: all at once)
        pre = F(np.sum(approx[:b].astype(F), dtype=F)) if b else F(0)
        e_bits = int(F(pre).view(np.uint32)) & 0x7f800000
        res = block_steps(xs[b * block:(b + 1) * block], e_bits) if e_bits else None
        summaries.append((e_bits, res))
    s, exact = F(0), 0
    for b in range(nblk):
        sbits = int(np.float32(s).view(np.uint32))
        e_bits = sbits & 0x7f800000
        pe, res = summaries[b]
        if res is not None and e_bits and pe == e_bits:
            total, has_tie, z_first = res
            S = (sbits & 0x7fffff) | 0x800000
            if has_tie and (S & 1):
                total += -1 if z_first else 1
            if S + total < (1 << 24):
                s = np.uint32(e_bits | ((S + total) & 0x7fffff)).view(F)
                continue
        exact += 1
        s = model_sum(xs[b * block:(b + 1) * block], block, s)      # k_fsum_sc16's step for this one block
    return s, exact, nblk


@pytest.mark.parametrize("name,xs", list(_cases()), ids=[c[0] for c in _cases()])
def test_wide_form_equals_sequential_float_sum(name, xs):
    want = sequential(xs)
    for rel_err in (1e-4, 0.3):                            # a prediction as good as the magnitudes give it / a poor one: speed, never the result
        got, exact, nblk = model_sum_wide(xs, 512, rel_err)
        assert got.view(np.uint32) == want.view(np.uint32), (name, rel_err, float(got), float(want))
    got, exact, nblk = model_sum_wide(xs, 512, 1e-4)
    if name.startswith("magsq small integers") or name == "sc16 magsq":
        assert exact <= 26 + nblk // 8, (name, exact, nblk)         # the binade crossings and the first block, not much more
