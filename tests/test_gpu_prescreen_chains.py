"""-m gpu: the pre-screen passes by chains (kernels/prescreen.inc: count_unit_chains / write_unit_chains — a round handles the
current segment of all four record chains of a unit) against the passes they replace, which follow one chain after the other.

The experiments build can run either pair (MGPU_PRESCREEN_VARIANT: 0 = the older passes, 3 = the product's) and, with bit 2, two
checking kernels: what the count pass left in the segment headers (live masks, a chain's live records before each segment, the
chains' totals) and what the write pass put into the live list, both against the chains walked one record after the other on the
device.  They print one line per difference; the write pass's first version lost records when the compiler carried part of a wave
into the next round on its own (the wave barrier at the end of a round, see the comment there) — this is the test that sees it.
Each case runs in a subprocess: library and switch are read from the environment."""
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
nfix, rate, dense, seconds, seed, chunk_buffers = {params!r}
iq = helpers.synth(seconds=seconds, seed=seed, rate=rate, dense=dense, threads=16)
want, wst = helpers.oracle_run(iq, 0, nfix, 1, 58)
d = readsb_amd.Demodulator(startup_time_ms=helpers.STARTUP_MS, max_samples=max(len(iq) // 2, 131072), nfix_crc=nfix, chunk_buffers=chunk_buffers or 0)
got, cnt = d.demodulate_capture(iq)
d.close()
helpers.assert_same_messages(got, want)
helpers.assert_same_counters(cnt, wst)
print("MESSAGES", len(got))
"""


def _run(params, variant):
    exp = os.path.join(helpers.ROOT, "readsb_amd", "csrc", "libmodes_gpu_exp.so")
    if not os.path.exists(exp):
        r = subprocess.run(["make", "-s", "-C", os.path.dirname(exp), "exp"], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
    env = dict(os.environ, MGPU_LIBRARY="libmodes_gpu_exp.so", MGPU_PRESCREEN_VARIANT=str(variant))
    code = SCRIPT.format(root=helpers.ROOT, tests=os.path.join(helpers.ROOT, "tests"), params=params)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-4000:]
    return r.stdout.splitlines()


@pytest.mark.parametrize("nfix,rate,dense,seconds,seed,chunk_buffers", [
    (1, 2000.0, 0, 20.0, 61, 64),        # several small chunks
    (2, 8000.0, 1, 20.0, 62, None),      # overlapping bursts, 2-bit repair: long chains, full segments
    (1, 3000.0, 0, 70.0, 63, None),      # an expiry of the filter inside the capture
])
def test_chain_passes_against_the_sequential_walk_of_the_chains(built, nfix, rate, dense, seconds, seed, chunk_buffers):
    out = _run((nfix, rate, dense, seconds, seed, chunk_buffers), 7)
    bad = [x for x in out if x.startswith("check:") or (x.startswith("live:") and "checked" not in x)]
    assert not bad, "\n".join(bad[:12])
    assert any(x.startswith("live: checked") for x in out), "the checking kernels did not run"
    assert any(x.startswith("MESSAGES") for x in out)


def test_older_passes_still_agree(built):
    """variant 0 (one chain after the other, both passes) and 1 (count by chains, write the older way) on the same capture."""
    n = [[x for x in _run((1, 2500.0, 0, 15.0, 64, 64), v) if x.startswith("MESSAGES")][-1] for v in (0, 1, 3)]
    assert n[0] == n[1] == n[2], n
