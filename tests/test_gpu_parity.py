"""-m gpu: the HIP path (through the C ABI) against the CPU oracle on the same seeded captures — on the BASELINE configurations
(test_uc8_options, test_uc8_10s_config2, test_dense_bursts) against the reference's own objects (oracle/_ref) where they are present.
Bit-exact bar: every accepted message (timestamp, frame bytes raw and corrected, score,
corrected bits, address, signal level) and every demod counter."""
import numpy as np
import pytest

import helpers

pytestmark = pytest.mark.gpu


def _demod(iq, **kw):
    import readsb_amd
    kw.setdefault("startup_time_ms", helpers.STARTUP_MS)
    kw.setdefault("max_samples", 64 * 131072)
    d = readsb_amd.Demodulator(**kw)
    try:
        msgs, counters = d.demodulate_capture(iq)
        return msgs, counters, d.timing()
    finally:
        d.close()


@pytest.mark.parametrize("nfix,fixdf,thr", [(1, 1, 58), (2, 1, 58), (0, 1, 58), (1, 0, 58), (1, 1, 75), (2, 1, 40)])
def test_uc8_options(built, nfix, fixdf, thr):
    iq = helpers.synth(seconds=3.0, seed=101)
    want, wst = helpers.reference_run(iq, 0, nfix, fixdf, thr)         # the reference's own objects (oracle/_ref) where they travel
    got, cnt, _ = _demod(iq, nfix_crc=nfix, fix_df=fixdf, preamble_threshold=thr)
    assert len(want) > 1000
    helpers.assert_same_messages(got, want)
    helpers.assert_same_counters(cnt, wst)


@pytest.mark.parametrize("thr", [40, 58, 400])
def test_clipped_capture_at_the_threshold_range_ends(built, thr):
    """The reference's whole --preamble-threshold range (40 .. 400, demod_2400.h:28-34 / readsb.c:1473) on a capture whose strong
    frames clip: runs of magnitude 65535 go through k_sweep's biased 16-bit dot products (32767 after the bias, the largest
    accumulators the tests can reach) and through the slicer's."""
    iq = helpers.synth(seconds=2.0, seed=2024, rate=3000.0)
    hot = np.clip((iq.astype(np.float32) - 127.5) * 2.6 + 127.5, 0, 255).round().astype(np.uint8)   # amplitude 40..119 LSB -> most frames saturate
    want, wst = helpers.reference_run(hot, 0, 1, 1, thr)
    lut = helpers.oracle_convert(hot[: 2 * 400000], 0)[0]
    assert (lut == 65535).mean() > 0.002 and len(want) > 200
    got, cnt, _ = _demod(hot, nfix_crc=1, fix_df=1, preamble_threshold=thr)
    helpers.assert_same_messages(got, want)
    helpers.assert_same_counters(cnt, wst)


def test_uc8_10s_config2(built):
    """BASELINE config 2: single 10 s UC8 stream, --fix."""
    iq = helpers.synth(seconds=10.0, seed=88172645463325252)
    want, wst = helpers.reference_run(iq)
    got, cnt, tm = _demod(iq)
    helpers.assert_same_messages(got, want)
    helpers.assert_same_counters(cnt, wst)
    assert tm["n_messages"] > 0


def test_chunked_feeds_equal_single_feed(built):
    """Feeding the stream in several calls (tail + filter state carried) changes nothing."""
    iq = helpers.synth(seconds=4.0, seed=5)
    want, wst = helpers.oracle_run(iq)
    import readsb_amd
    d = readsb_amd.Demodulator(startup_time_ms=helpers.STARTUP_MS, max_samples=8 * 131072)
    got, cnt = d.demodulate_capture(iq, chunk_samples=3 * 131072)
    d.close()
    helpers.assert_same_messages(got, want)
    helpers.assert_same_counters(cnt, wst)


def test_dense_bursts(built):
    """Config 5 flavour: overlapping 112-bit DF17 frames at 8000 msg/s."""
    iq = helpers.synth(seconds=2.0, seed=5, rate=8000.0, dense=1)
    want, wst = helpers.reference_run(iq, 0, 2, 1, 58)
    got, cnt, _ = _demod(iq, nfix_crc=2)
    helpers.assert_same_messages(got, want)
    helpers.assert_same_counters(cnt, wst)


@pytest.mark.parametrize("nsamples", [0, 1, 100, 326, 327, 4095, 4096, 131071, 131072, 131073, 2 * 131072, 300000])
def test_edge_lengths(built, nsamples):
    """Empty, tiny, exactly-one-buffer (extra zero-length buffer at EOF) and ragged captures."""
    iq = helpers.synth(nsamples=nsamples, seed=3, rate=4000.0)
    want, wst = helpers.oracle_run(iq)
    got, cnt, _ = _demod(iq)
    helpers.assert_same_messages(got, want)
    helpers.assert_same_counters(cnt, wst)


def test_noise_only(built):
    iq = helpers.synth(seconds=1.0, seed=9, rate=0.0)
    want, wst = helpers.oracle_run(iq)
    got, cnt, _ = _demod(iq)
    helpers.assert_same_messages(got, want)
    helpers.assert_same_counters(cnt, wst)


def test_caller_message_buffer(built):
    """mgpu_set_message_buffer: the same messages, built straight into the caller's array; too small an array
    fails loudly instead of dropping messages."""
    import readsb_amd
    iq = helpers.synth(seconds=3.0, seed=77)
    want, wst = helpers.oracle_run(iq)
    d = readsb_amd.Demodulator(startup_time_ms=helpers.STARTUP_MS, max_samples=64 * 131072)
    try:
        buf = np.empty(len(want) + 100, dtype=readsb_amd.MSG_DTYPE)
        d.set_message_buffer(buf)
        d.feed_iq(iq)
        d.finish()
        got, cnt = d.collect(out=buf)
        assert got.ctypes.data == buf.ctypes.data            # no copy: a view of the caller's array
        helpers.assert_same_messages(got, want)
        helpers.assert_same_counters(cnt, wst)
        small = np.empty(10, dtype=readsb_amd.MSG_DTYPE)
        d.reset()
        d.set_message_buffer(small)
        with pytest.raises(readsb_amd.MgpuError):
            d.feed_iq(iq)
    finally:
        d.close()


def test_long_chunk_walked_in_rounds(built):
    """One pipeline chunk of 2301 buffers (mgpu_config.chunk_buffers above the library's 1024): the kernels take the chunk in one
    launch each, the host's ordered walk takes it in rounds of about 1024 buffers (api.cpp: host_walk) — same messages, same counters."""
    B = 131072
    iq = helpers.synth(nsamples=2300 * B + 1000, seed=77, rate=2500.0)
    want, wst = helpers.reference_run(iq)
    got, cnt, tm = _demod(iq, max_samples=2304 * B, chunk_buffers=4096)
    assert len(want) > 100000
    helpers.assert_same_messages(got, want)
    helpers.assert_same_counters(cnt, wst)
