"""-m gpu: k_convert_* against the oracle's restatement of convert.c (bit-exact magnitudes)."""
import numpy as np
import pytest

import helpers

pytestmark = pytest.mark.gpu


def _dem(fmt, n):
    import readsb_amd
    return readsb_amd.Demodulator(fmt=fmt, max_samples=max(n, 131072))


def test_uc8_all_pairs(built):
    """Every (I, Q) byte pair, in an order that also exercises the unaligned chunk edges."""
    i, q = np.meshgrid(np.arange(256, dtype=np.uint8), np.arange(256, dtype=np.uint8), indexing="ij")
    iq = np.stack([i.ravel(), q.ravel()], axis=1).ravel()
    rng = np.random.default_rng(1)
    iq = np.concatenate([iq, rng.integers(0, 256, size=2 * 100003, dtype=np.uint8)])
    d = _dem(0, iq.size // 2)
    mag, ml, mp = d.convert(iq)
    d.close()
    want, wml, wmp = helpers.oracle_convert(iq, 0)
    assert np.array_equal(mag, want)
    assert ml == wml and mp == wmp      # integer sums => identical doubles


@pytest.mark.parametrize("n", [1, 5, 6, 7, 8, 9, 15, 333, 4096, 131072, 131073])
def test_uc8_ragged_lengths(built, n):
    rng = np.random.default_rng(n)
    iq = rng.integers(0, 256, size=2 * n, dtype=np.uint8)
    d = _dem(0, n)
    mag, ml, mp = d.convert(iq)
    d.close()
    want, wml, wmp = helpers.oracle_convert(iq, 0)
    assert np.array_equal(mag, want) and ml == wml and mp == wmp


def test_sc16q11_all_12bit_pairs(built):
    """Exhaustive over the 12-bit signed range bladeRF produces: 4096 x 4096 pairs."""
    v = np.arange(-2048, 2048, dtype=np.int16)
    i, q = np.meshgrid(v, v, indexing="ij")
    iq = np.stack([i.ravel(), q.ravel()], axis=1).ravel().astype("<i2")
    d = _dem(2, iq.size // 2)
    mag, ml, mp = d.convert(iq)
    d.close()
    want, wml, wmp = helpers.oracle_convert(iq, 2)
    assert np.array_equal(mag, want)
    # The reference accumulates mean level / power in a FLOAT running sum (convert.c:342-366): over 16.7 M samples that sum stops
    # absorbing small addends (it reads 0.779 where the true mean is 0.738).  The device reproduces that sum, rounding for rounding.
    exact_ml = float(np.minimum(np.sqrt((i.ravel().astype(np.float64) ** 2 + q.ravel().astype(np.float64) ** 2)) / 2048.0, 1.0).mean())
    assert ml == wml and mp == wmp
    assert abs(wml - exact_ml) < 0.1


@pytest.mark.parametrize("fmt", [1, 2])
def test_sc16_random_and_extremes(built, fmt):
    rng = np.random.default_rng(7 + fmt)
    n = 4_000_003
    iq = rng.integers(-32768, 32768, size=2 * n, dtype=np.int32).astype("<i2")
    iq[:8] = np.array([-32768, -32768, 32767, 32767, 0, 0, -1, 1], dtype="<i2")
    small = rng.integers(-3000, 3000, size=2 * (n // 2), dtype=np.int32).astype("<i2")
    iq[2 * (n - n // 2):] = small
    d = _dem(fmt, n)
    mag, ml, mp = d.convert(iq)
    want, wml, wmp = helpers.oracle_convert(iq, fmt)
    assert np.array_equal(mag, want)
    assert ml == wml and mp == wmp                      # the reference's sequential float sums (convert.c:225-249), rounding for rounding
    # ... and for buffer-sized calls with quiet starts, all-zero stretches, a clipped burst and an odd length
    for k, length in enumerate((131072, 131072, 100001, 7, 1)):
        blk = iq[2 * 262144 * k: 2 * 262144 * k + 2 * length].copy()
        if k == 1:
            blk[: 2 * 5000] = 0
            blk[2 * 60000: 2 * 60100] = 32767
        if k == 2:
            blk[:] = (blk.astype(np.int32) // 700).astype("<i2")     # a weak signal: the sum crawls through many binades
        m2, ml2, mp2 = d.convert(blk)
        w2, wml2, wmp2 = helpers.oracle_convert(blk, fmt)
        assert np.array_equal(m2, w2) and ml2 == wml2 and mp2 == wmp2, (k, ml2, wml2, mp2, wmp2)
    d.close()


@pytest.mark.parametrize("fmt", [1, 2])
@pytest.mark.parametrize("nstrong", [1, 2, 3])
def test_sc16_isolated_strong_samples_after_a_quiet_stretch(built, fmt, nstrong):
    """one to three full-scale samples in a 512-sample block while the running float sum is still small — each is
    2^22 grid steps and more of the binade the sum is in; the block has to be added in order, not as one integer addition."""
    rng = np.random.default_rng(90 + fmt + nstrong)
    n = 131072
    full = 32767 if fmt == 1 else 2047
    for quiet_amp, where in ((3, 700), (12, 5000), (40, 60000)):
        iq = rng.integers(-quiet_amp, quiet_amp + 1, size=2 * n, dtype=np.int32).astype("<i2")
        for j in range(nstrong):
            iq[2 * (where + 37 * j)] = full
            iq[2 * (where + 37 * j) + 1] = -full
        d = _dem(fmt, n)
        mag, ml, mp = d.convert(iq)
        d.close()
        want, wml, wmp = helpers.oracle_convert(iq, fmt)
        assert np.array_equal(mag, want)
        assert ml == wml and mp == wmp, (quiet_amp, where, ml, wml, mp, wmp)


# ---- the PIPELINE's magnitudes: since round 6 they come out of the fused sweep kernels (k_sweep_uc8, k_sweep_sc16), not out of the
#      converters mgpu_convert() runs — the same exhaustive inputs through a feed, read back with mgpu_debug_last_magnitudes ----
def _pipeline_magnitudes(fmt, iq, n):
    import readsb_amd
    d = readsb_amd.Demodulator(fmt=fmt, max_samples=max(n, 131072), startup_time_ms=helpers.STARTUP_MS)
    try:
        d.feed_iq(iq)
        fused = d.timing()["sweep_fused_chunks"]
        return d.last_magnitudes(n), fused
    finally:
        d.close()


def test_pipeline_magnitudes_uc8_all_pairs(built):
    i, q = np.meshgrid(np.arange(256, dtype=np.uint8), np.arange(256, dtype=np.uint8), indexing="ij")
    iq = np.stack([i.ravel(), q.ravel()], axis=1).ravel()
    rng = np.random.default_rng(1)
    iq = np.concatenate([iq, rng.integers(0, 256, size=2 * 100003, dtype=np.uint8)])     # ragged: the last steps are staged sample by sample
    n = iq.size // 2
    got, fused = _pipeline_magnitudes(0, iq, n)
    assert fused >= 1, "the feed did not go through k_sweep_uc8"
    assert np.array_equal(got, helpers.oracle_convert(iq, 0)[0])


@pytest.mark.parametrize("n", [1, 7, 333, 1024, 1025, 131072, 131073, 3 * 131072 + 698])
def test_pipeline_magnitudes_uc8_ragged_lengths(built, n):
    rng = np.random.default_rng(n)
    iq = rng.integers(0, 256, size=2 * n, dtype=np.uint8)
    got, _ = _pipeline_magnitudes(0, iq, n)
    assert np.array_equal(got, helpers.oracle_convert(iq, 0)[0])


def test_pipeline_magnitudes_sc16q11_all_12bit_pairs(built):
    v = np.arange(-2048, 2048, dtype=np.int16)
    i, q = np.meshgrid(v, v, indexing="ij")
    iq = np.stack([i.ravel(), q.ravel()], axis=1).ravel().astype("<i2")
    n = iq.size // 2
    got, fused = _pipeline_magnitudes(2, iq, n)
    assert fused >= 1, "the feed did not go through k_sweep_sc16"
    assert np.array_equal(got, helpers.oracle_convert(iq, 2)[0])


@pytest.mark.parametrize("fmt", [1, 2])
def test_pipeline_magnitudes_sc16_random_and_extremes(built, fmt):
    rng = np.random.default_rng(7 + fmt)
    n = 2_000_003
    iq = rng.integers(-32768, 32768, size=2 * n, dtype=np.int32).astype("<i2")
    iq[:8] = np.array([-32768, -32768, 32767, 32767, 0, 0, -1, 1], dtype="<i2")
    iq[2 * (n - n // 2):] = rng.integers(-3000, 3000, size=2 * (n // 2), dtype=np.int32).astype("<i2")
    got, fused = _pipeline_magnitudes(fmt, iq, n)
    assert fused >= 1
    assert np.array_equal(got, helpers.oracle_convert(iq, fmt)[0])
