"""-m gpu: the other input formats and the struct mag_buf entry point."""
import numpy as np
import pytest

import helpers

pytestmark = pytest.mark.gpu


def _run(iq, **kw):
    import readsb_amd
    kw.setdefault("startup_time_ms", helpers.STARTUP_MS)
    kw.setdefault("max_samples", 64 * 131072)
    d = readsb_amd.Demodulator(**kw)
    try:
        return d.demodulate_capture(iq)
    finally:
        d.close()


def _float_sum_counters_ok(cnt, wst):
    """SC16 paths: everything integer must match; noise power uses the reference's order-dependent
    float running sums (convert.c:225-249) and is compared with a tolerance."""
    for f in helpers.COUNTER_FIELDS:
        assert (np.asarray(cnt[f], dtype=np.uint64) == np.asarray(wst[f], dtype=np.uint64)).all(), f
    assert cnt["signal_power_sum"] == float(wst["signal_power_sum"])
    assert cnt["peak_signal_power"] == float(wst["peak_signal_power"])
    a, b = cnt["noise_power_sum"], float(wst["noise_power_sum"])
    assert abs(a - b) <= 2e-2 * abs(b)


def test_sc16q11_aggressive_config3(built):
    """BASELINE config 3: SC16Q11 stream, --aggressive (2-bit syndrome table)."""
    iq = helpers.synth(seconds=5.0, seed=7, fmt=2, rate=2500.0)
    want, wst = helpers.oracle_run(iq, 2, 2, 1, 58)
    got, cnt = _run(iq, fmt=2, nfix_crc=2)
    assert wst["demod_accepted"][2] > 20
    helpers.assert_same_messages(got, want)
    _float_sum_counters_ok(cnt, wst)


def test_sc16(built):
    iq = helpers.synth(seconds=3.0, seed=9, fmt=1)
    want, wst = helpers.oracle_run(iq, 1, 1, 1, 58)
    got, cnt = _run(iq, fmt=1)
    helpers.assert_same_messages(got, want)
    _float_sum_counters_ok(cnt, wst)


def test_mag_buf_entry(built):
    """demodulate2400(struct mag_buf *) replacement: magnitudes come from the host (here: the
    oracle's converter), one 131072-sample buffer with its 326-sample overlap per call."""
    import readsb_amd
    B, TR = 131072, 326
    iq = helpers.synth(nsamples=5 * B + 70000, seed=21)
    want, wst, mag = helpers.oracle_run(iq, want_mag=True)
    n = iq.size // 2
    d = readsb_amd.Demodulator(startup_time_ms=helpers.STARTUP_MS, max_samples=B)
    k = 0
    while True:
        length = min(B, n - k * B)
        data = mag[k * B: k * B + TR + length]
        new = data[TR:].astype(np.uint64)
        mean_power = float((new * new).sum()) / 65535.0 / 65535.0 / length if length else float("nan")
        st = k * B * 5
        d.demod_mag_buf(data, length, st, st // 12000 + helpers.STARTUP_MS, mean_power)
        k += 1
        if length < B:
            break
    got, cnt = d.collect()
    d.close()
    helpers.assert_same_messages(got, want)
    helpers.assert_same_counters(cnt, wst)


def test_two_minute_stream_filter_expiry(built):
    """130 s: the ICAO filter flips twice; aircraft fall silent and are forgotten."""
    iq = helpers.synth(seconds=130.0, seed=11, rate=1500.0, naircraft=300)
    want, wst = helpers.oracle_run(iq)
    assert wst["nflips"] >= 3
    got, cnt = _run(iq, max_samples=600 * 131072)
    helpers.assert_same_messages(got, want)
    helpers.assert_same_counters(cnt, wst)
