"""-m gpu: BASELINE configs[4] in C (readsb_amd/host/readsb_gpu_shard.c) — one capture time-chunked over N ranks, every rank walking
and building its own range through the protocol of include/modes_gpu.h (pre-pass over the expiry windows, the pass through the
ordinary pipeline with the schedule imposed, rounds over schedule + seam states, the sequential double sums re-added block-wise on
rank 0).  One rank over RCCL (what a 1-GPU box allows), and two / three ranks sharing the GPU over the program's file transport
(every all-gather and the final gather as files): the beast stream must be the reference's wire format of the reference's messages,
the counters — the two order-dependent double sums as bit patterns — the reference's."""
import ctypes as C
import json
import os
import subprocess
import uuid

import numpy as np
import pytest

import helpers

pytestmark = pytest.mark.gpu
EXE = os.path.join(helpers.ROOT, "readsb_amd", "host", "readsb_gpu_shard")


def _want_stream(want):
    lib = helpers.oracle_lib()
    lib.modes_oracle_beast_frame.restype = C.c_size_t
    lib.modes_oracle_beast_frame.argtypes = [C.c_void_p, C.c_void_p]
    frame, stream = (C.c_uint8 * 64)(), bytearray()
    for k in range(len(want)):
        stream += bytes(frame[: lib.modes_oracle_beast_frame(want[k:k + 1].ctypes.data, frame)])
    return bytes(stream)


def _check(line, want, wst):
    got = json.loads(line)
    assert got["messages"] == len(want)
    for f in helpers.COUNTER_FIELDS:
        assert np.asarray(got[f], dtype=np.uint64).tolist() == np.asarray(wst[f], dtype=np.uint64).tolist(), f
    for f in ("signal_power_sum", "peak_signal_power"):
        assert np.uint64(got[f + "_bits"]).view(np.float64) == float(wst[f]), f
    a, b = np.uint64(got["noise_power_sum_bits"]).view(np.float64), float(wst["noise_power_sum"])
    assert (np.isnan(a) and np.isnan(b)) or a == b
    return got


@pytest.mark.parametrize("world,seconds,rate,dense,opts", [(1, 6.0, 3000.0, 0, ["--aggressive"]), (2, 9.0, 5000.0, 1, []),
                                                            (3, 200.0, 1500.0, 0, [])])       # 200 s over 3 ranks: expiries, a warm-up that does not reach the start
def test_c_shard_gives_the_reference_stream_and_counters(built, tmp_path, world, seconds, rate, dense, opts):
    iq = helpers.synth(seconds=seconds, seed=7000 + world, rate=rate, dense=dense, threads=16)
    n = iq.size // 2
    if world == 2:
        iq = iq[: (n - n % 131072) * 2]                          # a whole number of buffers: the EOF buffer's clock, NaN and lost samples
    path = tmp_path / "cap.iq"
    iq.tofile(path)
    nfix = 2 if "--aggressive" in opts else 1
    want, wst = helpers.reference_run(iq, 0, nfix, 1, 58)
    assert len(want) > 3000
    base = [EXE, "--world", str(world), "--ifile", str(path), "--startup-time-ms", str(helpers.STARTUP_MS)] + opts
    out = tmp_path / "beast.bin"
    if world == 1:
        r = subprocess.run(base + ["--rank", "0", "--id-file", str(tmp_path / "nccl.id"), "--run-id", uuid.uuid4().hex[:12], "--out", str(out)], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-3000:]
        line = r.stdout.strip().splitlines()[-1]
    else:
        tdir = tmp_path / "xfer"
        tdir.mkdir()
        run_id = uuid.uuid4().hex[:12]                       # (the same on every rank, new for every run: stale files of another run are never read)
        procs = [subprocess.Popen(base + ["--rank", str(k), "--file-transport", str(tdir), "--run-id", run_id] + (["--out", str(out)] if k == 0 else []),
                                  stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for k in range(world)]
        outs = [p.communicate(timeout=900) for p in procs]
        for k, p in enumerate(procs):
            assert p.returncode == 0, f"rank {k}: " + outs[k][0][-1000:] + outs[k][1][-3000:]
        line = outs[0][0].strip().splitlines()[-1]
    got = _check(line, want, wst)
    assert got["ranks"] == world and got["rounds"] >= 1 and got["sum_blocks"] >= 1
    assert out.read_bytes() == _want_stream(want)
