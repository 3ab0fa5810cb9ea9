"""-m gpu: the experiments build of the library (`make -C readsb_amd/csrc exp` -> libmodes_gpu_exp.so) carries alternative forms of
product kernels that were measured and not kept; the ones that stay in the tree stay exact — run in a subprocess because library
and switches are read from the environment when the context is created.  (Generation 3's fused k_sweep_slice, cross-checked here
until round 4, is gone from the tree in round 5.)"""
import os
import subprocess
import sys

import pytest

import helpers

pytestmark = pytest.mark.gpu

WIDE_SCRIPT = r"""
import sys
sys.path.insert(0, {root!r}); sys.path.insert(0, {tests!r})
import numpy as np
import helpers, readsb_amd
for fmt, mode_ac in ((2, 0), (1, 0), (2, 1)):
    iq = helpers.synth(seconds=1.5, seed=606 + fmt, rate=2500.0, fmt=fmt)
    iq = np.ascontiguousarray(iq).view("<i2").copy()
    iq[2 * 200000: 2 * 200003] = 32767 if fmt == 1 else 2047            # isolated strong samples after a quiet stretch, inside a stream
    want, wst = helpers.oracle_run(iq, fmt, 2, 1, 58, mode_ac=mode_ac)
    d = readsb_amd.Demodulator(fmt=fmt, nfix_crc=2, mode_ac=mode_ac, startup_time_ms=helpers.STARTUP_MS, max_samples=64 * 131072)
    got, cnt = d.demodulate_capture(iq)
    d.close()
    helpers.assert_same_messages(got, want)
    helpers.assert_same_counters(cnt, wst)                             # float_tol 0.0: the noise statistics are made of the float sums
print("OK")
"""


def test_wide_float_sums_are_exact(built):
    """MGPU_FSUM_WIDE=1 (experiments build): k_fsum_approx / k_fsum_prep / k_fsum_apply in place of k_fsum_sc16 — the same sequential
    float sums, bit for bit (Mode S noise statistics, Mode A/C's noise floor)."""
    exp = os.path.join(helpers.ROOT, "readsb_amd", "csrc", "libmodes_gpu_exp.so")
    env = dict(os.environ, MGPU_FSUM_WIDE="1", MGPU_LIBRARY="libmodes_gpu_exp.so")
    code = WIDE_SCRIPT.format(root=helpers.ROOT, tests=os.path.join(helpers.ROOT, "tests"))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert r.stdout.strip().endswith("OK")


@pytest.mark.parametrize("buffers,ragged,dense", [(16, 0, 0), (5, 1025, 1), (3, 7 * 1024 + 1, 0), (1, 0, 1)])
def test_sweep_kernel_alone_matches_cpu_scan(built, buffers, ragged, dense):
    """k_sweep on its own (tools/micro/sweep_cold.hip includes the product's kernels.hip): the candidate lists of a chunk — an
    even and an odd number of 1024-position steps, fewer steps than resident waves, a ragged end — against a plain CPU scan of
    the same magnitudes (pre-check + threshold tests, demod_2400.c:311-378)."""
    import json
    exe = os.path.join(helpers.ROOT, "tools", "micro", "sweep_cold")
    if not os.path.exists(exe):
        r = subprocess.run(["make", "-s", "-C", os.path.join(helpers.ROOT, "tools"), "micro"], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
    r = subprocess.run([exe, str(buffers), "2", "2", str(dense), "2000", "0", str(ragged)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["mismatches_vs_cpu_scan"] == 0 and out["candidates"] == out["candidates_cpu"] > 0


@pytest.mark.parametrize("buffers,dense", [(16, 0), (3, 1), (1, 0)])
def test_fused_sweep_kernel_alone_matches_cpu(built, buffers, dense):
    """k_sweep_uc8 on its own (tools/micro/sweep_uc8_cold.hip includes the product's kernels.hip and launches through the library's
    launch_sweep): every magnitude against init_uc8_lookup's table (convert.c:35-62), every buffer's exact sum(mag) / sum(mag^2)
    (convert.c:64-108) and the candidate lists against a plain CPU scan (demod_2400.c:311-378) — more steps than resident waves,
    fewer, and a single buffer."""
    import json
    exe = os.path.join(helpers.ROOT, "tools", "micro", "sweep_uc8_cold")
    if not os.path.exists(exe):
        r = subprocess.run(["make", "-s", "-C", os.path.join(helpers.ROOT, "tools"), "micro"], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
    r = subprocess.run([exe, str(buffers), "2", "3", str(dense)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["magnitude_mismatches_vs_cpu_table"] == 0 and out["buffer_sum_mismatches"] == 0 and out["candidate_mismatches_vs_cpu_scan"] == 0
    assert out["candidates"] == out["candidates_cpu"] > 0


FUSED_SCRIPT = r"""
import sys
sys.path.insert(0, {root!r}); sys.path.insert(0, {tests!r})
import numpy as np
import helpers, readsb_amd
out = []
for buf_samples, nsamples in ((4096, 700000), (8192, 83 * 8192 + 5000), (32768, 40 * 32768 + 1234), (131072, 9 * 131072 + 77)):
    iq = helpers.synth(nsamples=nsamples, seed=4242 + buf_samples, rate=3000.0)
    d = readsb_amd.Demodulator(startup_time_ms=helpers.STARTUP_MS, max_samples=16 * 131072, buf_samples=buf_samples)
    msgs, cnt = d.demodulate_capture(iq)
    tm = d.timing()
    d.close()
    out.append((buf_samples, len(msgs), msgs.tobytes(), {{k: (v.tolist() if hasattr(v, "tolist") else v) for k, v in cnt.items()}}, tm["sweep_fused_chunks"]))
import pickle
sys.stdout.buffer.write(pickle.dumps(out))
"""


def test_fused_sweep_equals_the_two_kernels_at_every_buffer_size(built):
    """k_sweep_uc8 (converter and sweep in one pass, the product for UC8) against k_convert_uc8_lean + k_sweep (experiments build,
    MGPU_SWEEP_FUSED=0) where the oracle cannot follow: buffers of 4096 .. 131072 samples (mgpu_config.buf_samples; the oracle's grid
    is the reference's 131072).  A buffer of 4096 samples is four steps: every fourth step closes a buffer's exact sum(mag) /
    sum(mag^2) (its first 326 positions still belong to the buffer before) — messages, timestamps and every counter, the noise
    power made of those sums included, must be the same bytes."""
    import pickle
    code = FUSED_SCRIPT.format(root=helpers.ROOT, tests=os.path.join(helpers.ROOT, "tests"))
    res = {}
    for fused in ("1", "0"):
        env = dict(os.environ, MGPU_SWEEP_FUSED=fused, MGPU_LIBRARY="libmodes_gpu_exp.so")
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, timeout=600)
        assert r.returncode == 0, r.stderr[-4000:].decode(errors="replace")
        res[fused] = pickle.loads(r.stdout)
    for a, b in zip(res["1"], res["0"]):
        assert a[4] > 0 and b[4] == 0, "the switch did not select the kernels"
        assert a[1] == b[1] > 50, (a[0], a[1], b[1])
        assert a[2] == b[2], f"messages differ at buf_samples {a[0]}"
        assert repr(a[3]) == repr(b[3]), f"counters differ at buf_samples {a[0]}: {[(k, a[3][k], b[3][k]) for k in a[3] if repr(a[3][k]) != repr(b[3][k])]}"
        assert a[3]["noise_power_sum"] == a[3]["noise_power_sum"] > 0, "the captures end inside a buffer: the noise power is a number"
