"""-m gpu: the C host side (readsb_amd/host): `readsb_gpu_ifile` = `readsb --device-type ifile
--raw --mlat` with the demodulator on the GPU, compared line by line with what the reference prints
for the same file (displayModesMessage in --raw --mlat mode, mode_s.c:1834-1847)."""
import os
import subprocess

import pytest

import helpers

pytestmark = pytest.mark.gpu
CLI = os.path.join(helpers.ROOT, "readsb_amd", "host", "readsb_gpu_ifile")


def _expected_lines(msgs):
    out = []
    for m in msgs:
        out.append("@%012X%s;" % (int(m["timestamp"]), bytes(m["msg"][: int(m["msgbits"]) // 8]).hex()))
    return out


@pytest.mark.parametrize("fmt,flag,nfix", [(0, "--fix", 1), (2, "--aggressive", 2)])
def test_cli_matches_reference_raw_output(built, tmp_path, fmt, flag, nfix):
    iq = helpers.synth(seconds=3.0, seed=77, fmt=fmt)
    path = tmp_path / "cap.iq"
    iq.tofile(path)
    want, wst = helpers.oracle_run(iq, fmt, nfix, 1, 58)
    r = subprocess.run([CLI, "--device-type", "ifile", "--ifile", str(path), "--iformat", helpers.FMT_NAMES[fmt], flag,
                        "--raw", "--mlat", "--stats", "--gpu-chunk-buffers", "7"], capture_output=True, text=True, check=True)
    got = r.stdout.strip().splitlines()
    assert got == _expected_lines(want)
    assert f"{int(wst['demod_preambles'])} Mode-S message preambles received" in r.stderr
    assert f"{int(wst['demod_accepted'][0])} accepted with correct CRC" in r.stderr
