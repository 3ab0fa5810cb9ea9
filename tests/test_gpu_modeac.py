"""-m gpu: Mode A/C replies (demodulate2400AC, demod_2400.c:575-761) beside Mode S, in netUseMessage order
(per buffer: the Mode S messages, then the replies), against the CPU oracle; UC8 input (the reply threshold
depends on the converter's exact per-buffer sums)."""
import numpy as np
import pytest

import helpers

pytestmark = pytest.mark.gpu


def _run(iq, **kw):
    import readsb_amd
    d = readsb_amd.Demodulator(startup_time_ms=helpers.STARTUP_MS, max_samples=max(len(iq) // 2, 131072), mode_ac=1, **kw)
    try:
        return d.demodulate_capture(iq)
    finally:
        d.close()


@pytest.mark.parametrize("seconds,rate,seed,nfix", [(3.0, 800.0, 404, 1), (2.0, 300.0, 405, 2), (60.0, 700.0, 406, 1)])
def test_mode_ac_beside_mode_s(built, seconds, rate, seed, nfix):
    iq = helpers.synth(seconds=seconds, seed=seed, rate=rate, dense=2, threads=16)
    want, wst = helpers.oracle_run(iq, 0, nfix, 1, 58, mode_ac=1)
    got, cnt = _run(iq, nfix_crc=nfix)
    nac = int((want["msgtype"] == 77).sum())
    assert nac > 200 and (want["msgtype"] != 77).sum() > 200
    helpers.assert_same_messages(got, want)
    helpers.assert_same_counters(cnt, wst)
    assert int(cnt["demod_modeac"]) == nac == int(wst["demod_modeac"])


def test_mode_ac_off_is_the_default(built):
    import readsb_amd
    iq = helpers.synth(seconds=2.0, seed=404, rate=800.0, dense=2)
    want, wst = helpers.oracle_run(iq)                      # no Mode A/C
    d = readsb_amd.Demodulator(startup_time_ms=helpers.STARTUP_MS, max_samples=len(iq) // 2)
    got, cnt = d.demodulate_capture(iq)
    d.close()
    helpers.assert_same_messages(got, want)
    assert int(cnt["demod_modeac"]) == 0
