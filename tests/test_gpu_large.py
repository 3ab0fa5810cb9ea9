"""-m gpu: multi-chunk captures (several 512-buffer pipeline chunks, filter flips, the parallel walk with
batch restarts) against the CPU oracle, in the configurations the benchmark does not run: SC16Q11 with
2-bit repair (BASELINE config 3 flavour), dense overlapping bursts (config 5 flavour), no repair at all."""
import pytest

import helpers

pytestmark = pytest.mark.gpu


def _run(iq, **kw):
    import readsb_amd
    n = len(iq) // helpers.FMT_BYTES[kw.get("fmt", 0)]
    d = readsb_amd.Demodulator(startup_time_ms=helpers.STARTUP_MS, max_samples=max(n, 131072), **kw)
    try:
        return d.demodulate_capture(iq)
    finally:
        d.close()


@pytest.mark.parametrize("fmt,nfix,fixdf,thr,rate,dense,seconds,seed", [
    (2, 2, 1, 58, 2000.0, 0, 70.0, 31),      # SC16Q11, --aggressive: float converter, 2-bit syndrome tables (20 KB of LDS keys)
    (0, 2, 1, 58, 8000.0, 1, 60.0, 32),      # UC8, overlapping DF17 bursts at 8000 msg/s
    (0, 0, 0, 75, 3000.0, 0, 100.0, 33),     # UC8, no CRC repair, raised threshold, > 60 s: filter expiry inside the capture
])
def test_multi_chunk_capture(built, fmt, nfix, fixdf, thr, rate, dense, seconds, seed):
    iq = helpers.synth(seconds=seconds, fmt=fmt, seed=seed, rate=rate, dense=dense, threads=16)
    want, wst = helpers.oracle_run(iq, fmt, nfix, fixdf, thr)
    got, cnt = _run(iq, fmt=fmt, nfix_crc=nfix, fix_df=fixdf, preamble_threshold=thr)
    assert len(want) > 50000
    helpers.assert_same_messages(got, want)
    helpers.assert_same_counters(cnt, wst)          # (SC16 formats included: their mean power is the reference's float sum, bit for bit)
