"""CPU: the host's ordered walk (readsb_amd/csrc/resolve.cpp) on a REAL chunk — the live records, signal powers and buffer
clocks of one 512-buffer chunk of the benchmark stream, dumped from a GPU run (MGPU_DUMP_DIR) and committed as
tests/golden/walk_chunk.tgz.  The exact speculative parallel walk (ranges of buffers walked at once, committed in stream
order) must make the serial walk's decisions, with 2, 4 and 7 ranges, from a cold filter and from a warm one; the final
filter contents must agree.  (tests/test_cabi.py covers the same on synthetic streams through mgpu_selftest_walk; the
sanitizer builds of this replay are in tools/sanitize_host.sh.)"""
import os
import re
import subprocess
import tarfile

import pytest

import helpers


@pytest.fixture(scope="module")
def replay(tmp_path_factory):
    d = tmp_path_factory.mktemp("walk")
    with tarfile.open(os.path.join(helpers.GOLDEN_DIR, "walk_chunk.tgz")) as t:
        t.extractall(d)
    exe = str(d / "walk_replay")
    subprocess.run(["g++", "-O2", "-std=c++17", "-I/opt/rocm/include", "-D__HIP_PLATFORM_AMD__", "-pthread", "-o", exe,
                    os.path.join(helpers.TOOLS_DIR, "walk_replay.cpp"), os.path.join(helpers.ROOT, "readsb_amd", "csrc", "resolve.cpp")], check=True)
    return exe, str(d)


@pytest.mark.parametrize("ranges", [2, 4, 7])
def test_parallel_walk_equals_serial_walk_on_a_real_chunk(replay, ranges):
    exe, d = replay
    r = subprocess.run([exe, d, str(ranges)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.strip().splitlines()
    assert re.match(r"48752 records, 512 buffers -> 26915 messages", lines[0]), lines[0]
    rounds = [ln for ln in lines if ln.startswith("round")]
    assert len(rounds) == 4
    for ln in rounds:
        assert "identical 1, filter unions equal 1" in ln, ln
        m = re.search(r"serial (\d+) msgs, parallel\(\d+\) (\d+) msgs", ln)
        assert m and m.group(1) == m.group(2)
    assert "serial 26915 msgs" in rounds[0] and "serial 27010 msgs" in rounds[1]      # cold filter, then warm


def test_device_walk_model_equals_serial_walk_on_a_real_chunk(replay):
    """The device walk's algorithm (kernels/walk.inc restated in Resolver::device_walk_model) on the same real chunk: the cold
    round is the one the library hands to the host (the filter table grows while the aircraft are learnt), the warm ones are
    decided by the model, in at most three walks — always with the serial walk's decisions and final filter."""
    exe, d = replay
    r = subprocess.run([exe, d, "2"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    rounds = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("model round")]
    assert len(rounds) == 4, r.stdout
    for ln in rounds:
        assert "identical 1" in ln, ln
    assert "decided by the model 0" in rounds[0]
    assert all("decided by the model 1" in ln for ln in rounds[1:]), rounds
    assert all(int(re.search(r"walks (\d+)", ln).group(1)) <= 3 for ln in rounds)
