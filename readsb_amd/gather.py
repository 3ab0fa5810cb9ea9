"""Aggregator step of the multi-GPU job: every rank demodulates its own stream (or its own
time-chunk of one capture), then the decoded message counts and the fixed-size 64-byte message
records are gathered on rank 0 — the role TCP beast forwarding between readsb daemons plays in
the reference (`--net-connector`, README.md:40-51; net_io.c:1655).  No collective is needed
inside the data path: buffers are independent.

Backend-agnostic (`nccl` = RCCL over xGMI on the MI355X node, `gloo` in the CPU tests)."""
import numpy as np
import torch
import torch.distributed as dist


def gather_messages(msgs: np.ndarray, device: torch.device, dst: int = 0):
    """msgs: structured array of 64-byte mgpu_msg records of THIS rank.
    Returns (counts list for every rank, list of per-rank record arrays on dst / None elsewhere)."""
    world, rank = dist.get_world_size(), dist.get_rank()
    rec = msgs.dtype.itemsize
    cnt = torch.tensor([len(msgs)], dtype=torch.int64, device=device)
    counts = [torch.zeros_like(cnt) for _ in range(world)]
    dist.all_gather(counts, cnt)                       # 8 bytes per rank
    counts = [int(c.item()) for c in counts]
    cap = max(counts) if counts else 0
    buf = torch.zeros(max(cap, 1) * rec, dtype=torch.uint8, device=device)
    if len(msgs):
        flat = torch.from_numpy(np.ascontiguousarray(msgs).view(np.uint8).reshape(-1))
        buf[: flat.numel()] = flat.to(device)
    gathered = [torch.zeros_like(buf) for _ in range(world)] if rank == dst else None
    dist.gather(buf, gathered, dst=dst)                # padded fixed-size records
    if rank != dst:
        return counts, None
    out = []
    for r in range(world):
        raw = gathered[r][: counts[r] * rec].cpu().numpy()
        out.append(raw.view(msgs.dtype).copy())
    return counts, out


class _DeviceBytes:
    """nbytes of device memory at a raw address, as something torch.as_tensor accepts without copying."""

    def __init__(self, ptr: int, nbytes: int):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}


class MessageGatherer:
    """The same aggregator step without a host round trip and off the critical path.

    Per rank: a ring of pinned staging buffers the demodulator's messages are collected INTO
    (`staging()`), one asynchronous H2D copy, an 8-byte `all_gather` of counts and a fixed-size `gather`
    of the records to `dst` — all issued with `async_op=True`, so step k's exchange runs over xGMI
    while step k+1 demodulates.  The gathered records stay in `dst`'s HBM (`wait()` returns device
    tensors): that is where an on-GPU beast encoder / aggregator picks them up; `fetch()` copies them
    to the host for tests.  Shapes are static (`capacity` records per rank) so nothing in `submit()`
    waits for a count.  Backend-agnostic like gather_messages (`gloo` + CPU tensors in the tests)."""

    def __init__(self, dtype, device: torch.device, capacity: int, dst: int = 0, depth: int = 2, host_alloc=None):
        """host_alloc (optional): nbytes -> page-locked uint8 numpy array, e.g. Demodulator.host_alloc, which places the staging
        buffers on the GPU's NUMA node (torch's own pinned allocator places them where the calling thread happens to run)."""
        self.dtype, self.device, self.capacity, self.dst, self.depth = dtype, device, int(capacity), dst, depth
        self.world, self.rank = dist.get_world_size(), dist.get_rank()
        self.rec = dtype.itemsize
        # one trailer record behind the capacity: the in-place device path (device_buffer / submit_inplace) sends the count there,
        # inside the records' own collective
        nbytes = (self.capacity + 1) * self.rec
        cuda = device.type == "cuda"
        if host_alloc is not None and cuda:
            self.host = [torch.from_numpy(host_alloc(nbytes)) for _ in range(depth)]
        else:
            self.host = [torch.empty(nbytes, dtype=torch.uint8, pin_memory=cuda) for _ in range(depth)]
        self.dev = [torch.empty(nbytes, dtype=torch.uint8, device=device) for _ in range(depth)] if cuda else self.host
        self.recv = ([[torch.empty(nbytes, dtype=torch.uint8, device=device) for _ in range(self.world)] for _ in range(depth)]
                     if self.rank == dst else [None] * depth)
        self.cnt_host = [torch.zeros(1, dtype=torch.int64, pin_memory=cuda) for _ in range(depth)]
        self.cnt = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(depth)]
        self.counts = [[torch.zeros(1, dtype=torch.int64, device=device) for _ in range(self.world)] for _ in range(depth)]
        self.pending = [None] * depth
        self.copied = [torch.cuda.Event() if cuda else None for _ in range(depth)]   # H2D of the slot has read the staging buffer
        # everything the gatherer enqueues goes on a stream of its own: torch's default stream is the legacy null stream, and
        # work on it is ordered against the rest of the device's queues by the runtime
        self.stream = torch.cuda.Stream(device=device) if cuda else None
        self.seq = 0

    def staging(self, ahead: int = 0) -> np.ndarray:
        """Record array (capacity entries) of the next slot (ahead=1: of the one after it — a demodulator with deferred
        feeds fills step k+1's array before step k's is submitted; needs depth >= 3); safe to overwrite once this returns."""
        if not 0 <= ahead < self.depth - 1 and ahead != 0:
            raise ValueError("staging(ahead): the ring is too short")
        k = (self.seq + ahead) % self.depth
        self._wait_slot(k)
        return self.host[k].numpy().view(self.dtype)

    def submit(self, n: int) -> int:
        """The first n records of staging() are this rank's messages of the step.  Returns the slot."""
        if n > self.capacity:
            raise ValueError(f"{n} messages exceed the gatherer's capacity of {self.capacity}")
        k = self.seq % self.depth
        self._wait_slot(k)
        self.cnt_host[k][0] = n
        with self._on_stream():
            return self._submit_host(k, n)

    def _on_stream(self):
        import contextlib
        return torch.cuda.stream(self.stream) if self.stream is not None else contextlib.nullcontext()

    def _submit_host(self, k, n):
        self.cnt[k].copy_(self.cnt_host[k], non_blocking=True)
        import os as _os
        _dbg = _os.environ.get("MGPU_DBG_GATHER", "")
        if self.dev is not self.host and n and "nocopy" not in _dbg:
            self.dev[k][: n * self.rec].copy_(self.host[k][: n * self.rec], non_blocking=True)
        if self.copied[k] is not None:
            self.copied[k].record()
        if "nocoll" in _dbg:
            self.pending[k] = ()
        else:
            w1 = dist.all_gather(self.counts[k], self.cnt[k], async_op=True)
            w2 = dist.gather(self.dev[k], self.recv[k], dst=self.dst, async_op=True)
            self.pending[k] = (w1, w2)
        self.seq += 1
        return k

    def submit_device(self, dptr: int, n: int) -> int:
        """The step's n records already are in this rank's HBM at device address dptr (Demodulator.collect_feed_device): no
        staging buffer, no upload — one device-to-device copy into the slot's fixed-size gather buffer, then the exchange."""
        if n > self.capacity:
            raise ValueError(f"{n} messages exceed the gatherer's capacity of {self.capacity}")
        k = self.seq % self.depth
        self._wait_slot(k)
        self.cnt_host[k][0] = n
        with self._on_stream():
            return self._submit_device(k, dptr, n)

    def _submit_device(self, k, dptr, n):
        self.cnt[k].copy_(self.cnt_host[k], non_blocking=True)
        import os as _os
        if n and "nocopy" not in _os.environ.get("MGPU_DBG_GATHER", ""):
            src = torch.as_tensor(_DeviceBytes(dptr, n * self.rec), device=self.device)
            self.dev[k][: n * self.rec].copy_(src, non_blocking=True)
        if self.copied[k] is not None:
            self.copied[k].record()
        if "nocoll" in _os.environ.get("MGPU_DBG_GATHER", ""):
            self.pending[k] = ()
            self.seq += 1
            return k
        w1 = dist.all_gather(self.counts[k], self.cnt[k], async_op=True)
        w2 = dist.gather(self.dev[k], self.recv[k], dst=self.dst, async_op=True)
        self.pending[k] = (w1, w2)
        self.seq += 1
        return k

    # ---- the demodulator writes its records straight into the slot's gather buffer (mgpu_set_device_message_buffer) ----
    def device_buffer(self, ahead: int = 0):
        """(device address, capacity in records) of the next slot's gather buffer (ahead=1: of the one after it), for
        Demodulator.set_device_message_buffer: k_build_messages writes the feed's records where the collective reads them — no
        device-to-device copy (18.8 MB per feed, measured 0.10 ms of every 1.35 ms feed as torch's copy beside the pipeline's
        kernels: profiles/r06_gather_vs_plain.txt).  Safe to overwrite once this returns."""
        if not 0 <= ahead < self.depth - 1 and ahead != 0:
            raise ValueError("device_buffer(ahead): the ring is too short")
        k = (self.seq + ahead) % self.depth
        self._wait_slot(k)
        return int(self.dev[k].data_ptr()), self.capacity

    def submit_inplace(self, n: int) -> int:
        """The next slot's buffer holds this rank's n records of the step (the demodulator wrote them there).  ONE collective: the
        count rides in the buffer's trailer record (the 8-byte all_gather of counts beside every gather was a second launch and a
        second synchronisation among the ranks per feed)."""
        if n > self.capacity:
            raise ValueError(f"{n} messages exceed the gatherer's capacity of {self.capacity}")
        k = self.seq % self.depth
        self._wait_slot(k)
        self.cnt_host[k][0] = n
        with self._on_stream():
            self.dev[k][self.capacity * self.rec: self.capacity * self.rec + 8].view(torch.int64).copy_(self.cnt_host[k], non_blocking=True)
            if self.copied[k] is not None:
                self.copied[k].record()
            self.pending[k] = (dist.gather(self.dev[k], self.recv[k], dst=self.dst, async_op=True),)
            self.inplace = True
        self.seq += 1
        return k

    def _wait_slot(self, k):
        if self.pending[k] is not None:
            with self._on_stream():
                for w in self.pending[k]:
                    w.wait()                  # (NCCL: orders the current stream — the gatherer's own — after the collective)
            if self.copied[k] is not None:
                self.copied[k].synchronize()  # the host may overwrite the staging buffer again
            self.pending[k] = None

    def wait(self, k=None):
        """Complete slot k (default: everything outstanding).  Returns (counts, per-rank record tensors on
        `dst` — None elsewhere) of the most recently submitted slot (or of slot k)."""
        slots = range(self.depth) if k is None else [k]
        for i in slots:
            self._wait_slot(i)
        if self.stream is not None:
            self.stream.synchronize()
        last = (self.seq - 1) % self.depth if k is None else k
        if getattr(self, "inplace", False):      # the counts came in the trailer records (dst only; the other ranks know their own)
            if self.recv[last] is None:
                return [int(self.cnt_host[last][0]) if r == self.rank else -1 for r in range(self.world)], None
            o = self.capacity * self.rec
            return [int(self.recv[last][r][o: o + 8].view(torch.int64).item()) for r in range(self.world)], self.recv[last]
        counts = [int(c.item()) for c in self.counts[last]]
        return counts, self.recv[last]

    def fetch(self, k=None):
        """Host copies of what wait() returned: (counts, list of per-rank record arrays on dst / None)."""
        counts, recv = self.wait(k)
        if recv is None:
            return counts, None
        return counts, [recv[r][: counts[r] * self.rec].cpu().numpy().view(self.dtype).copy() for r in range(self.world)]


def merge_by_timestamp(per_rank):
    """Rank-0 side: one list ordered by the 12 MHz timestamp (stable for equal stamps)."""
    allm = np.concatenate(per_rank) if per_rank else np.zeros(0)
    order = np.argsort(allm["timestamp"], kind="stable")
    return allm[order]
