"""-m gpu: the ordered accept walk on the device (kernels/walk.inc, MGPU_DEVICE_WALK).

check mode runs it beside the host walk of every chunk — same records, same filter state — and compares every decision
(record, buffer, score, the window-statistics inputs), every counter, and the filter state after
Resolver::apply_device_walk with the state the host walk left: mgpu_debug_device_walk()[6] must stay 0.  The streams are
the ones the host walk is tested with (against the CPU oracle): aircraft that appear during the capture (adds), the 60 s
expiry inside a chunk, dense overlapping bursts, small chunks (many chunk boundaries)."""
import pytest

import helpers

pytestmark = pytest.mark.gpu

B = 131072


def _run(iq, monkeypatch, mode, chunk_buffers=None, **kw):
    import readsb_amd
    monkeypatch.setenv("MGPU_DEVICE_WALK", mode)
    if chunk_buffers:
        monkeypatch.setenv("MGPU_CHUNK_BUFFERS", str(chunk_buffers))
    n = len(iq) // helpers.FMT_BYTES[kw.get("fmt", 0)]
    d = readsb_amd.Demodulator(startup_time_ms=helpers.STARTUP_MS, max_samples=max(n, B), **kw)
    try:
        got, cnt = d.demodulate_capture(iq)
        return got, cnt, d.device_walk_stats()
    finally:
        d.close()


@pytest.mark.parametrize("nfix,rate,dense,seconds,seed,chunk_buffers,serial", [
    (1, 2000.0, 0, 30.0, 41, 64, False),     # several small chunks: the aircraft population is learnt in the first ones (the table grows)
    (2, 8000.0, 1, 20.0, 42, None, False),   # overlapping DF17 bursts
    (1, 3000.0, 0, 130.0, 43, None, False),  # two expiries inside the capture
    (1, 3000.0, 0, 70.0, 44, 128, True),     # every buffer through the serial decision loop (the fallback of the lane-parallel one)
])
def test_device_walk_equals_host_walk(built, monkeypatch, nfix, rate, dense, seconds, seed, chunk_buffers, serial):
    if serial:
        monkeypatch.setenv("MGPU_DBG_WK_SERIAL", "1")
    iq = helpers.synth(seconds=seconds, seed=seed, rate=rate, dense=dense, threads=16)
    want, wst = helpers.oracle_run(iq, 0, nfix, 1, 58)
    got, cnt, st = _run(iq, monkeypatch, "check", chunk_buffers, nfix_crc=nfix)
    helpers.assert_same_messages(got, want)
    helpers.assert_same_counters(cnt, wst)
    print("device walk:", st)
    assert st["differences"] == 0
    assert st["chunks"] >= 2 and st["device"] >= st["chunks"] // 2, st


@pytest.mark.parametrize("nfix,rate,dense,seconds,seed,chunk_buffers", [
    (1, 2000.0, 0, 30.0, 51, 64),
    (2, 8000.0, 1, 20.0, 52, None),
    (1, 3000.0, 0, 130.0, 53, None),
])
def test_device_walk_results_equal_the_oracle(built, monkeypatch, nfix, rate, dense, seconds, seed, chunk_buffers):
    """MGPU_DEVICE_WALK=1: the decisions come from the device, the records never reach the host."""
    iq = helpers.synth(seconds=seconds, seed=seed, rate=rate, dense=dense, threads=16)
    want, wst = helpers.oracle_run(iq, 0, nfix, 1, 58)
    got, cnt, st = _run(iq, monkeypatch, "1", chunk_buffers, nfix_crc=nfix)
    print("device walk:", st)
    helpers.assert_same_messages(got, want)
    helpers.assert_same_counters(cnt, wst)
    assert st["device"] >= st["chunks"] // 2, st
