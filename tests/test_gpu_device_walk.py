"""-m gpu: the ordered accept walk on the device (kernels/walk.inc) — since round 4 in the EXPERIMENTS build only
(`make -C readsb_amd/csrc exp` -> libmodes_gpu_exp.so, MGPU_DEVICE_WALK): the product library has one walk, the host's
(DESIGN.md §3 says why).  Each case runs in a subprocess, because library and switches are read from the environment.

check mode runs the device walk beside the host walk of every chunk — same records, same filter state — and compares every
decision (record, buffer, score, the window-statistics inputs), every counter, and the filter state after
Resolver::apply_device_walk with the state the host walk left: mgpu_debug_device_walk()[6] must stay 0.  The streams are
the ones the host walk is tested with (against the CPU oracle): aircraft that appear during the capture (adds), the 60 s
expiry inside a chunk, dense overlapping bursts, small chunks (many chunk boundaries)."""
import json
import os
import subprocess
import sys

import pytest

import helpers

pytestmark = pytest.mark.gpu

SCRIPT = r"""
import json, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, {tests!r})
import helpers, readsb_amd
nfix, rate, dense, seconds, seed, chunk_buffers = {params!r}
iq = helpers.synth(seconds=seconds, seed=seed, rate=rate, dense=dense, threads=16)
want, wst = helpers.oracle_run(iq, 0, nfix, 1, 58)
d = readsb_amd.Demodulator(startup_time_ms=helpers.STARTUP_MS, max_samples=max(len(iq) // 2, 131072), nfix_crc=nfix, chunk_buffers=chunk_buffers or 0)
got, cnt = d.demodulate_capture(iq)
st = d.device_walk_stats()
d.close()
helpers.assert_same_messages(got, want)
helpers.assert_same_counters(cnt, wst)
print("STATS " + json.dumps(st))
"""


def _run(params, mode, serial=False):
    exp = os.path.join(helpers.ROOT, "readsb_amd", "csrc", "libmodes_gpu_exp.so")
    if not os.path.exists(exp):
        r = subprocess.run(["make", "-s", "-C", os.path.dirname(exp), "exp"], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
    env = dict(os.environ, MGPU_LIBRARY="libmodes_gpu_exp.so", MGPU_DEVICE_WALK=mode)
    if serial:
        env["MGPU_DBG_WK_SERIAL"] = "1"
    code = SCRIPT.format(root=helpers.ROOT, tests=os.path.join(helpers.ROOT, "tests"), params=params)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [x for x in r.stdout.splitlines() if x.startswith("STATS ")][-1]
    return json.loads(line[6:])


@pytest.mark.parametrize("nfix,rate,dense,seconds,seed,chunk_buffers,serial", [
    (1, 2000.0, 0, 30.0, 41, 64, False),     # several small chunks: the aircraft population is learnt in the first ones (the table grows)
    (2, 8000.0, 1, 20.0, 42, None, False),   # overlapping DF17 bursts
    (1, 3000.0, 0, 130.0, 43, None, False),  # two expiries inside the capture
    (1, 3000.0, 0, 70.0, 44, 128, True),     # every buffer through the serial decision loop (the fallback of the lane-parallel one)
])
def test_device_walk_equals_host_walk(built, nfix, rate, dense, seconds, seed, chunk_buffers, serial):
    st = _run((nfix, rate, dense, seconds, seed, chunk_buffers), "check", serial)
    print("device walk:", st)
    assert st["differences"] == 0
    assert st["chunks"] >= 2 and st["device"] >= st["chunks"] // 2, st


@pytest.mark.parametrize("nfix,rate,dense,seconds,seed,chunk_buffers", [
    (1, 2000.0, 0, 30.0, 51, 64),
    (2, 8000.0, 1, 20.0, 52, None),
    (1, 3000.0, 0, 130.0, 53, None),
    (1, 2500.0, 0, 140.0, 54, 128),          # >= 16 chunks with two expiries among them
])
def test_device_walk_results_equal_the_oracle(built, nfix, rate, dense, seconds, seed, chunk_buffers):
    """MGPU_DEVICE_WALK=1: the decisions come from the device, the records never reach the host."""
    st = _run((nfix, rate, dense, seconds, seed, chunk_buffers), "1")
    print("device walk:", st)
    assert st["device"] >= st["chunks"] // 2, st
