"""CPU test of the N>1 path: world_size-2 `gloo` run of the aggregator gather bench.py uses
over RCCL (readsb_amd/gather.py)."""
import os
import sys

import numpy as np
import torch.multiprocessing as mp

import helpers


def _worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, helpers.ROOT)
    from readsb_amd.binding import MSG_DTYPE
    from readsb_amd.gather import gather_messages, merge_by_timestamp
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(100 + rank)
    n = [700, 123][rank]                                  # ragged counts
    msgs = np.zeros(n, dtype=MSG_DTYPE)
    msgs["timestamp"] = np.sort(rng.integers(0, 10**9, size=n))
    msgs["addr"] = rank
    msgs["msg"] = rng.integers(0, 256, size=(n, 14), dtype=np.uint8)
    counts, per_rank = gather_messages(msgs, torch.device("cpu"))
    ok = counts == [700, 123]
    if rank == 0:
        ok = ok and len(per_rank) == 2 and per_rank[0].tobytes() == msgs.tobytes() and len(per_rank[1]) == 123
        ok = ok and bool((per_rank[1]["addr"] == 1).all())
        merged = merge_by_timestamp(per_rank)
        ok = ok and len(merged) == 823 and bool((np.diff(merged["timestamp"]) >= 0).all())
    else:
        ok = ok and per_rank is None
    # empty rank edge case
    counts2, per2 = gather_messages(msgs[:0] if rank == 1 else msgs[:5], torch.device("cpu"))
    ok = ok and counts2 == [5, 0]
    if rank == 0:
        ok = ok and len(per2[1]) == 0 and len(per2[0]) == 5
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_gather_two_ranks_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert results == {0: True, 1: True}


def _worker_async(rank, world, port, q):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, helpers.ROOT)
    from readsb_amd.binding import MSG_DTYPE
    from readsb_amd.gather import MessageGatherer
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = MessageGatherer(MSG_DTYPE, torch.device("cpu"), capacity=1000)
    ok = True
    sent = []
    for step in range(5):                                  # more steps than ring slots: slots are reused
        n = [700 - 100 * step, 123 + step][rank]
        out = g.staging()
        out[:n]["timestamp"] = np.arange(n) + 1000 * step + 7 * rank
        out[:n]["addr"] = rank * 16 + step
        sent.append(out[:n].copy())
        slot = g.submit(n)
        if step == 2:                                      # look at a step while later ones are still to come
            counts, per_rank = g.fetch(slot)
            ok = ok and counts == [500, 125]
            if rank == 0:
                ok = ok and per_rank[0].tobytes() == sent[2].tobytes() and bool((per_rank[1]["addr"] == 16 + 2).all())
            else:
                ok = ok and per_rank is None
    counts, per_rank = g.fetch()
    ok = ok and counts == [300, 127]
    if rank == 0:
        ok = ok and per_rank[0].tobytes() == sent[4].tobytes() and len(per_rank[1]) == 127
        ok = ok and bool((per_rank[1]["timestamp"] == np.arange(127) + 4000 + 7).all())
    try:
        g.submit(1001)
        ok = False
    except ValueError:
        pass
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_async_gatherer_two_ranks_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker_async, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert results == {0: True, 1: True}


def _worker_inplace_ahead(rank, world, port, q):
    """bench.py's N > 1 protocol: a ring of A + 2 slots, the records of up to A feeds written into the slots ahead of the one being
    gathered (device_buffer(ahead)), one collective per feed with the count in the slot's trailer record (submit_inplace)."""
    import ctypes
    import torch
    import torch.distributed as dist
    sys.path.insert(0, helpers.ROOT)
    from readsb_amd.binding import MSG_DTYPE
    from readsb_amd.gather import MessageGatherer
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    A, cap = 2, 900
    g = MessageGatherer(MSG_DTYPE, torch.device("cpu"), capacity=cap, depth=A + 2)
    ok = True
    sent = {}

    def fill(k):                                           # what mgpu_set_device_message_buffer + the feed's build do on the GPU
        ptr, c = g.device_buffer(ahead=k - g.seq)
        n = [600 - 50 * k, 100 + k][rank]
        view = np.frombuffer((ctypes.c_uint8 * (c * MSG_DTYPE.itemsize)).from_address(ptr), dtype=MSG_DTYPE)
        view[:n]["timestamp"] = np.arange(n) + 1000 * k + 7 * rank
        view[:n]["addr"] = rank * 64 + k
        sent[k] = (n, view[:n].copy())

    count = 7
    for k in range(count + A):
        if k < count:
            fill(k)
        if k >= A:
            j = k - A
            slot = g.submit_inplace(sent[j][0])
            if j in (3, count - 1):
                counts, per_rank = g.fetch(slot)
                if rank == 0:
                    ok = ok and counts == [600 - 50 * j, 100 + j]
                    ok = ok and per_rank[0].tobytes() == sent[j][1].tobytes() and bool((per_rank[1]["addr"] == 64 + j).all()) and len(per_rank[1]) == 100 + j
                else:
                    ok = ok and per_rank is None and counts[rank] == sent[j][0]
    try:
        g.device_buffer(ahead=A + 1)
        ok = False
    except ValueError:
        pass
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_inplace_gather_with_feeds_ahead_two_ranks_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker_inplace_ahead, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert results == {0: True, 1: True}
