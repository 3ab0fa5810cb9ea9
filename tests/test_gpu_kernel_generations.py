"""-m gpu: the cross-check build of the library (`make -C readsb_amd/csrc exp` -> libmodes_gpu_exp.so) carries the superseded
fused kernel k_sweep_slice (generation 3) next to the shipped pair k_sweep + k_slice (5); both must be bit-identical to the
oracle — run in a subprocess because library and generation are read from the environment when the context is created."""
import os
import subprocess
import sys

import pytest

import helpers

pytestmark = pytest.mark.gpu

SCRIPT = r"""
import sys
sys.path.insert(0, {root!r}); sys.path.insert(0, {tests!r})
import helpers, readsb_amd
iq = helpers.synth(seconds=3.0, seed=404, rate=3000.0)
want, wst = helpers.oracle_run(iq, 0, 2, 1, 58)
d = readsb_amd.Demodulator(nfix_crc=2, startup_time_ms=helpers.STARTUP_MS, max_samples=64 * 131072)
got, cnt = d.demodulate_capture(iq)
helpers.assert_same_messages(got, want)
helpers.assert_same_counters(cnt, wst)
print("OK", len(got))
"""


@pytest.mark.parametrize("version", ["3", "5"])
def test_generation_matches_oracle(built, version):
    exp = os.path.join(helpers.ROOT, "readsb_amd", "csrc", "libmodes_gpu_exp.so")
    if not os.path.exists(exp):
        r = subprocess.run(["make", "-s", "-C", os.path.dirname(exp), "exp"], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
    env = dict(os.environ, MGPU_SWEEP_VERSION=version, MGPU_LIBRARY="libmodes_gpu_exp.so")
    code = SCRIPT.format(root=helpers.ROOT, tests=os.path.join(helpers.ROOT, "tests"))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.strip().startswith("OK")


def test_fused_uc8_convert_matches_oracle(built):
    """MGPU_FUSED_CONVERT=1: convert_uc8_nodc inside k_sweep's tile load (k_sweep_t<true>: d_mag, the per-buffer sums and the
    candidates all come from one kernel) — ragged length, several chunks, 2-bit repair; messages and every counter as the oracle's."""
    script = SCRIPT.replace("seconds=3.0, seed=404", "nsamples=37 * 131072 + 4321, seed=405").replace("max_samples=64 * 131072", "max_samples=16 * 131072")
    env = dict(os.environ, MGPU_FUSED_CONVERT="1")
    code = script.format(root=helpers.ROOT, tests=os.path.join(helpers.ROOT, "tests"))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.strip().startswith("OK")
