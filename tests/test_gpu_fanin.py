"""-m gpu: the fan-in host row (readsb_amd/host/sdr_gpu_fanin.c, CLI readsb_gpu_fanin): several sample files of different
formats demodulated concurrently, one context per stream in one process — every stream's output must be what the
reference prints for that file alone (--raw --mlat lines, mode_s.c:1834-1847)."""
import os
import subprocess

import pytest

import helpers

pytestmark = pytest.mark.gpu
CLI = os.path.join(helpers.ROOT, "readsb_amd", "host", "readsb_gpu_fanin")


def _expected_lines(msgs):
    return ["@%012X%s;" % (int(m["timestamp"]), bytes(m["msg"][: int(m["msgbits"]) // 8]).hex()) for m in msgs]


def test_streams_are_independent_and_exact(built, tmp_path):
    specs = [(0, 3.0, 71, 1500.0), (1, 2.0, 72, 2500.0), (0, 0.3, 73, 900.0), (2, 2.5, 74, 2000.0), (0, 4.0, 75, 3000.0)]
    args, want = [], []
    for k, (fmt, seconds, seed, rate) in enumerate(specs):
        iq = helpers.synth(seconds=seconds, seed=seed, fmt=fmt, rate=rate)
        path = tmp_path / f"cap{k}.iq"
        iq.tofile(path)
        args += ["--iformat", helpers.FMT_NAMES[fmt], "--ifile", str(path)]
        msgs, st = helpers.oracle_run(iq, fmt, 1, 1, 58)
        want.append((_expected_lines(msgs), st))
    prefix = tmp_path / "out"
    r = subprocess.run([CLI] + args + ["--fix", "--out-prefix", str(prefix), "--stats", "--gpu-chunk-buffers", "5"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    for k, (lines, st) in enumerate(want):
        got = open(f"{prefix}.{k}").read().strip().splitlines()
        assert len(lines) > 100 and got == lines, f"stream {k}"
        assert f"stream {k} " in r.stderr and f"{int(st['demod_preambles'])} Mode-S message preambles received" in r.stderr
    assert "fan-in: 5 streams" in r.stderr


def test_fanin_refuses_without_input(built, tmp_path):
    r = subprocess.run([CLI, "--out-prefix", str(tmp_path / "x")], capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "requires an --ifile argument" in r.stderr
