"""-m gpu: BASELINE config 5 — one capture sharded by buffer ranges (readsb_amd/shard.py) must give the unsharded
message list AND every demodulator statistic bit for bit: dense overlapping bursts, several shard counts, shards that start mid-stream with
326 samples of history, a capture long enough for a shard's 120 s warm-up NOT to reach back to the start (aircraft that fell
silent and expired before it), and a two-rank gloo run (both ranks on the one GPU of the test box)."""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

import helpers

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("nshards,seconds,dense,rate,nfix", [(2, 8.0, 1, 8000.0, 2), (3, 8.0, 0, 2000.0, 1), (8, 70.0, 1, 6000.0, 1),
                                                             (2, 290.0, 0, 1500.0, 1)])     # second range from 145 s: warm-up [25 s, 145 s)
def test_sharded_capture_equals_unsharded(built, nshards, seconds, dense, rate, nfix):
    import readsb_amd
    from readsb_amd.shard import demodulate_sharded_local
    from readsb_amd import shard
    iq = helpers.synth(seconds=seconds, seed=900 + nshards + int(seconds), rate=rate, dense=dense, threads=16)
    if seconds > 200:
        assert shard.warmup_start(shard.shard_ranges(iq.size // 2, nshards)[-1][0]) > 0     # the cutoff is exercised
    want, wst = helpers.oracle_run(iq, 0, nfix, 1, 58)
    d = readsb_amd.Demodulator(nfix_crc=nfix, startup_time_ms=helpers.STARTUP_MS, max_samples=256 * 131072)
    try:
        got, cnt = demodulate_sharded_local(d, iq, nshards)
    finally:
        d.close()
    assert len(want) > 5000
    helpers.assert_same_messages(got, want)
    helpers.assert_same_counters(cnt, wst)          # every demodulator statistic, the ones the skip windows and the samples feed included


@pytest.mark.parametrize("nshards,seconds,dense,rate,nfix,naircraft", [(2, 8.0, 1, 8000.0, 2, 200), (3, 8.0, 0, 2000.0, 1, 200), (8, 70.0, 1, 6000.0, 1, 200),
                                                                       (2, 290.0, 0, 1500.0, 1, 200),      # second range from 145 s: its warm-up [24.7 s, 145 s) starts from an empty filter
                                                                       (3, 400.0, 0, 1200.0, 1, 3000)])    # many aircraft: the table grows early, the later ranks' warm-ups must find its size
def test_sharded_walk_equals_unsharded(built, nshards, seconds, dense, rate, nfix, naircraft):
    """Round 4: every rank walks and builds its OWN range (shard.py: ShardWalkRank, run_walk_protocol) — the unsharded message
    list and every counter, the order-dependent double sums included, bit for bit."""
    import readsb_amd
    from readsb_amd import shard
    iq = helpers.synth(seconds=seconds, seed=1900 + nshards + int(seconds), rate=rate, dense=dense, naircraft=naircraft, threads=16)
    want, wst = helpers.ref_run(iq, 0, nfix, 1, 58) if helpers.have_ref() else helpers.oracle_run(iq, 0, nfix, 1, 58)
    d = readsb_amd.Demodulator(nfix_crc=nfix, startup_time_ms=helpers.STARTUP_MS, max_samples=256 * 131072)
    stats = {}
    try:
        got, cnt = shard.demodulate_sharded_walk_local(d, iq, nshards, stats)
    finally:
        d.close()
    assert len(want) > 5000
    helpers.assert_same_messages(got, want)
    helpers.assert_same_counters(cnt, wst)
    assert stats["rounds"] >= 1 and min(stats["walks"]) >= 1 and stats["sum_blocks"] >= 1
    if seconds < 100:
        assert stats["rounds"] == 1 and not any(stats["imported"])  # every warm-up reaches back to the start: one walk per rank


@pytest.mark.parametrize("nshards,seconds,dense,rate,nfix,naircraft,deferred", [(2, 8.0, 1, 8000.0, 2, 200, False), (8, 70.0, 1, 6000.0, 1, 200, False),
                                                                                (2, 290.0, 0, 1500.0, 1, 200, False), (3, 400.0, 0, 1200.0, 1, 3000, False),
                                                                                # round 5: warm-up and range as deferred feeds, the walker marks the range's begin itself
                                                                                (8, 70.0, 1, 6000.0, 1, 200, True), (2, 290.0, 0, 1500.0, 1, 200, True),
                                                                                (3, 400.0, 0, 1200.0, 1, 3000, True)])
def test_sharded_stream_equals_unsharded(built, nshards, seconds, dense, rate, nfix, naircraft, deferred):
    """... and the form whose walk and build overlap the kernels: the schedule from a pre-pass over the buffers an expiry can follow
    (shard.expiry_windows), every rank's warm-up + range through the ordinary pipeline (mgpu_shard_stream_*)."""
    import readsb_amd
    from readsb_amd import shard
    iq = helpers.synth(seconds=seconds, seed=2900 + nshards + int(seconds), rate=rate, dense=dense, naircraft=naircraft, threads=16)
    want, wst = helpers.ref_run(iq, 0, nfix, 1, 58) if helpers.have_ref() else helpers.oracle_run(iq, 0, nfix, 1, 58)
    # (deferred: a feed call must hold a whole warm-up — up to 120 s + a buffer — or a whole range)
    d = readsb_amd.Demodulator(nfix_crc=nfix, startup_time_ms=helpers.STARTUP_MS, max_samples=(4096 if deferred else 256) * 131072)
    stats = {}
    try:
        got, cnt = shard.demodulate_sharded_stream_local(d, iq, nshards, stats, out_capacity=len(want) + 4096 if deferred else None)
    finally:
        d.close()
    assert len(want) > 5000
    helpers.assert_same_messages(got, want)
    helpers.assert_same_counters(cnt, wst)
    assert stats["rounds"] >= 1 and min(stats["walks"]) >= 1
    if seconds < 100:
        assert stats["rounds"] == 1 and not any(stats["imported"])


def _rank_walk(rank, world, port, q, seconds, seed, form="packets"):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, helpers.ROOT)
    sys.path.insert(0, os.path.join(helpers.ROOT, "tests"))
    import readsb_amd
    from readsb_amd.shard import demodulate_sharded_walk, demodulate_sharded_stream
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    iq = helpers.synth(seconds=seconds, seed=seed, rate=4000.0, dense=1, threads=8)
    d = readsb_amd.Demodulator(startup_time_ms=helpers.STARTUP_MS, max_samples=128 * 131072)
    stats = {}
    res = (demodulate_sharded_stream if form == "stream" else demodulate_sharded_walk)(d, iq, torch.device("cpu"), stats=stats)
    d.close()
    ok = True
    if rank == 0:
        want, wst = helpers.oracle_run(iq)
        got, cnt = res
        try:
            helpers.assert_same_messages(got, want)
            helpers.assert_same_counters(cnt, wst)
            ok = len(want) > 5000 and stats["rounds"] >= 1
        except AssertionError as e:
            sys.stderr.write(f"sharded walk, rank 0: {e}\n")
            ok = False
    else:
        ok = res is None
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("form", ["packets", "stream"])
def test_sharded_walk_two_ranks_gloo(built, form):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 36500 + (os.getpid() % 2000) + (7 if form == "stream" else 0)
    procs = [ctx.Process(target=_rank_walk, args=(r, 2, port, q, 6.0, 4243, form)) for r in range(2)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert results == {0: True, 1: True}


def _rank(rank, world, port, q, seconds, seed):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, helpers.ROOT)
    sys.path.insert(0, os.path.join(helpers.ROOT, "tests"))
    import readsb_amd
    from readsb_amd.shard import demodulate_sharded
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    iq = helpers.synth(seconds=seconds, seed=seed, rate=4000.0, dense=1, threads=8)
    d = readsb_amd.Demodulator(startup_time_ms=helpers.STARTUP_MS, max_samples=128 * 131072)
    res = demodulate_sharded(d, iq, torch.device("cpu"))
    d.close()
    ok = True
    if rank == 0:
        want, wst = helpers.oracle_run(iq)
        got, cnt = res
        try:
            helpers.assert_same_messages(got, want)
            helpers.assert_same_counters(cnt, wst)
            ok = len(want) > 5000
        except AssertionError:
            ok = False
    else:
        ok = res is None
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_capture_two_ranks_gloo(built):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_rank, args=(r, 2, port, q, 6.0, 4242)) for r in range(2)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert results == {0: True, 1: True}
