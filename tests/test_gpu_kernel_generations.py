"""-m gpu: the earlier generations of k_sweep_slice (MGPU_SWEEP_VERSION=1|2, kept for A/B timing)
must stay bit-identical to the oracle too — run in a subprocess because the version is read from
the environment when the context is created."""
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


@pytest.mark.parametrize("version", ["1", "2", "3", "4"])
def test_generation_matches_oracle(built, version):
    env = dict(os.environ, MGPU_SWEEP_VERSION=version)
    code = SCRIPT.format(root=helpers.ROOT, tests=os.path.join(helpers.ROOT, "tests"))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.strip().startswith("OK")
