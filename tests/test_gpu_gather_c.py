"""-m gpu: the aggregator exchange in C (readsb_amd/host/readsb_gpu_gather.c: demodulate -> ncclAllGather of counts -> records to
rank 0 over RCCL -> one beast stream encoded on rank 0's GPU), run with the one rank a 1-GPU box allows: the stream must be the
reference's wire format of the reference's messages (oracle messages through the restated modesSendBeastOutput).  The N > 1 form of
the same exchange is covered by tests/test_dist_gloo.py (Python / gloo) and bench.py's --dryrun-gloo run below."""
import ctypes as C
import json
import os
import subprocess
import sys

import pytest

import helpers

pytestmark = pytest.mark.gpu
EXE = os.path.join(helpers.ROOT, "readsb_amd", "host", "readsb_gpu_gather")


def test_c_gather_single_rank_writes_the_reference_beast_stream(built, tmp_path):
    iq = helpers.synth(seconds=4.0, seed=515, rate=3000.0)
    path, out, idf = tmp_path / "cap.iq", tmp_path / "beast.bin", tmp_path / "nccl.id"
    iq.tofile(path)
    want, _ = helpers.oracle_run(iq, 0, 2, 1, 58)
    r = subprocess.run([EXE, "--rank", "0", "--world", "1", "--id-file", str(idf), "--ifile", str(path), "--aggressive",
                        "--startup-time-ms", str(helpers.STARTUP_MS), "--out", str(out)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert f"{len(want)} messages gathered" in r.stderr
    lib = helpers.oracle_lib()
    lib.modes_oracle_beast_frame.restype = C.c_size_t
    lib.modes_oracle_beast_frame.argtypes = [C.c_void_p, C.c_void_p]
    frame, stream = (C.c_uint8 * 64)(), bytearray()
    for k in range(len(want)):
        stream += bytes(frame[: lib.modes_oracle_beast_frame(want[k:k + 1].ctypes.data, frame)])
    assert out.read_bytes() == bytes(stream)


def test_c_gather_forward_only_is_what_the_reference_program_forwards(built, tmp_path):
    """--forward-only: the gate and the gated encoder on the gathered records.  On the golden capture the stream + the deferred
    messages the whole reference program forwarded is that program's --dump-beast file."""
    import numpy as np
    import gate_util as gu
    name = "uc8_fix_2s"
    kw, opt = gu.CASES[name]
    iq = helpers.synth(threads=8, **kw)
    path, out, idf, dfile = tmp_path / "cap.iq", tmp_path / "beast.bin", tmp_path / "nccl.id", tmp_path / "deferred.bin"
    iq.tofile(path)
    r = subprocess.run([EXE, "--rank", "0", "--world", "1", "--id-file", str(idf), "--ifile", str(path), "--fix", "--forward-only",
                        "--deferred-out", str(dfile), "--startup-time-ms", str(helpers.STARTUP_MS), "--out", str(out)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "forwarded frames only" in r.stderr
    gold = open(os.path.join(helpers.GOLDEN_DIR, f"beast_{name}.bin"), "rb").read()
    fwd = gu.golden_forwarded(name)
    msgs, _ = helpers.oracle_run(iq, 0, opt["nfix"], 1, 58)
    deferred = np.fromfile(dfile, dtype="<u8").reshape(-1, 3)
    lib = helpers.oracle_lib()
    lib.modes_oracle_beast_frame.restype = C.c_size_t
    lib.modes_oracle_beast_frame.argtypes = [C.c_void_p, C.c_void_p]
    frame = (C.c_uint8 * 64)()
    stream, spliced, at = out.read_bytes(), bytearray(), 0
    for rank, index, offset in deferred:
        assert rank == 0
        spliced += stream[at:int(offset)]
        at = int(offset)
        if fwd[int(index)]:
            spliced += bytes(frame[: lib.modes_oracle_beast_frame(msgs[int(index):int(index) + 1].ctypes.data, frame)])
    spliced += stream[at:]
    assert bytes(spliced) == gold


def test_bench_multi_rank_path_dry_run(built):
    """bench.py's N > 1 path (one stream per rank, deferred feeds into the gatherer's staging ring, asynchronous gather, per-rank
    bit-identity check and CPU baseline, max-over-ranks timing) with two ranks sharing the one GPU: gloo, collectives on CPU tensors."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29517", os.path.join(helpers.ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--samples", str(96 * 131072), "--dryrun-gloo"], capture_output=True, text=True, timeout=600, env=env, cwd=helpers.ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["config"]["streams"] == 2
    assert d["cpu_baseline"]["cores"] == 2 and d["cpu_baseline"]["bit_identical_to_gpu"] and len(d["cpu_baseline"]["per_stream_msamples_s"]) == 2


def test_bench_config5_two_shards_dry_run(built):
    """bench.py --config 5 (one dense-burst capture time-chunked over the ranks, strong scaling) with two ranks sharing the one GPU
    (gloo, CPU tensors): bitmap exchange, record packets to rank 0, ordered walk there — and the in-run check against the reference."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29519", os.path.join(helpers.ROOT, "bench.py"), "--gpus", "2", "--config", "5", "--steps", "2",
                        "--warmup", "1", "--samples", str(96 * 131072), "--dryrun-gloo"], capture_output=True, text=True, timeout=600, env=env,
                       cwd=helpers.ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["shards"] == 2
    assert d["cpu_baseline"]["bit_identical_to_gpu"] and d["messages_per_step"] == d["cpu_baseline"]["messages"] > 1000
