"""One capture sharded by buffer ranges over several GPUs (BASELINE config 5).

Buffers of 131072 samples are independent except for the ICAO filter: the ordered walk needs the filter's
state, and the GPU pre-screen needs to know which addresses could be in the filter at all — every address some clean
DF17 / DF11-IID0 frame carried ("adder", mode_s.c:766-779) in the last TWO filter generations: an address is dropped by the
second expiry after its last add, at most 120 s of sample time later (icao_filter.c:96-130, readsb.c:1227-1231).

  rank 0    owns the first range and puts it through the ordinary pipeline (nothing precedes it): messages straight away;
  rank r>0  sweeps the WARMUP (120 s + one buffer) of samples before its range for their adders only (pass "1": convert, sweep,
            slice, no records kept), then its own range against that bitmap plus its own adders as it goes (pass "2"), and
            ships the surviving records of every chunk as a packet (~45 bytes per 1000 samples, all the sweep-side statistics
            included) to rank 0;
  rank 0    continues its stream with the packets in range order (mgpu_walk_packets: the ordered walk and the message build
            exactly as for its own chunks).

No collective but the gather of the packets.  The result is the unsharded message list and every counter, bit for bit.  With
one rank this IS the unsharded pipeline.  With N ranks the GPU phase takes (1 + 120 s / range) / N of the unsharded time, and the
walk + build of the other ranks' packets on rank 0 does not shrink (Amdahl: on dense bursts it is ~1.5x what the whole
unsharded pipeline takes per sample, because there it hides behind the GPU) — sharding that walk across the ranks
(`Resolver::parallel_walk` with ranks in place of threads) is what would make this scale.  (Round 2 swept every range twice and
OR-ed the adder bitmaps of the whole capture over the ranks: twice the GPU work, and nothing overlapped.)

`shard_ranges` / `run_shard_pass*` / `rank_packets` are the pieces; `demodulate_sharded_local` runs all shards in one process
(tests, single GPU); `demodulate_sharded` is the torch.distributed version (one rank = one shard; `gloo` with host tensors or
`nccl` with device tensors)."""
import numpy as np

from .binding import _FMT_BYTES

BUF = 131072
TRAILING = 326
SAMPLE_RATE = 2400000
FILTER_TTL_S = 60                       # MODES_ICAO_FILTER_TTL, readsb.h:315: one generation
WARMUP = (2 * FILTER_TTL_S * SAMPLE_RATE + BUF - 1) // BUF * BUF + BUF   # two generations of samples, whole buffers, one to spare


def shard_ranges(nsamples, nshards, buf=BUF):
    """Contiguous whole-buffer ranges [first, last) in samples, one per shard (possibly empty at the end)."""
    nbuf = (nsamples + buf - 1) // buf
    out = []
    for s in range(nshards):
        b0, b1 = nbuf * s // nshards, nbuf * (s + 1) // nshards
        out.append((min(b0 * buf, nsamples), min(b1 * buf, nsamples)))
    return out


def warmup_start(first):
    """First sample a rank whose range starts at `first` has to look at (for adders only)."""
    return max(0, first - WARMUP)


def needed_from(first):
    """First sample of the capture the rank needs in memory (warm-up + the 326 samples of history before it)."""
    return max(0, warmup_start(first) - TRAILING)


def _feed_host(d, iq, first, last, bps):
    cap = int(d.cfg.max_samples)
    cap -= cap % BUF
    off = first
    while off < last:
        k = min(cap, last - off)
        d.feed_iq(iq[off * bps:(off + k) * bps])
        off += k


def _feed_resident(d, first, last, bps, resident):
    base_first, base_ptr = resident
    cap = int(d.cfg.max_samples)
    cap -= cap % BUF
    off = first
    while off < last:
        k = min(cap, last - off)
        d.feed_resident(k, base_ptr + (off - base_first) * bps)
        off += k


def _history(iq, first, bps, history):
    if first == 0:
        return None
    return history if history is not None else iq[(first - TRAILING) * bps:first * bps]


def run_shard_pass(d, iq, first, last, mode, bitmap=None, resident=None, history=None):
    """mode 1 -> the adder bitmap of [first, last); mode 2 (given the bitmap of the samples before) -> its record packets.
    resident = (first sample held, device address of it): the samples are already in device memory (the benchmark's form:
    nothing but the 326 history samples crosses PCIe), `history` = their bytes when the rank does not hold them in host memory."""
    bps = _FMT_BYTES[d.fmt]
    d.reset()
    if mode == 2 and bitmap is not None:
        d.set_adder_bitmap(bitmap)
    if last > first:
        d.shard_begin(first, _history(iq, first, bps, history), mode)
        if resident is not None:
            _feed_resident(d, first, last, bps, resident)
        else:
            _feed_host(d, iq, first, last, bps)
    return d.adder_bitmap() if mode == 1 else d.shard_packets()


def rank_packets(d, iq, first, last, resident=None, histories=None, phases=None, lap=None):
    """What a rank r > 0 does: the warm-up's adders, then its range's packets (a view of the library's buffer)."""
    wf = warmup_start(first)
    h_warm, h_own = histories if histories is not None else (None, None)
    bitmap = run_shard_pass(d, iq, wf, first, 1, None, resident, h_warm) if first > wf else None
    if lap:
        lap("warmup_adders")
    pk = run_shard_pass(d, iq, first, last, 2, bitmap, resident, h_own)
    if lap:
        lap("range_packets")
    return pk


def run_first_range(d, iq, last, resident=None, out=None):
    """Rank 0's own range [0, last) through the ordinary pipeline; the context is left ready for mgpu_walk_packets."""
    d.reset()
    if out is not None:
        d.set_message_buffer(out)
    bps = _FMT_BYTES[d.fmt]
    if resident is not None:
        _feed_resident(d, 0, last, bps, resident)
    else:
        _feed_host(d, iq, 0, last, bps)


def demodulate_sharded_local(d, iq, nshards):
    """All shards one after the other on one Demodulator (no communication): the algorithm's reference run.
    (The later ranges first, then range 0 and the walk: one context plays every rank.)"""
    iq = np.ascontiguousarray(iq).view(np.uint8).reshape(-1)
    n = iq.size // _FMT_BYTES[d.fmt]
    ranges = shard_ranges(n, nshards)
    packets = [rank_packets(d, iq, first, last).copy() for first, last in ranges[1:]]   # (copies: the next pass reuses the buffer)
    run_first_range(d, iq, ranges[0][1])
    for pk in packets:
        if pk.size:
            d.walk_packets(pk)
    d.finish()
    return d.collect()


def demodulate_sharded(d, iq, device=None, resident=None, nsamples=None, histories=None, phases=None, out=None):
    """torch.distributed version: rank r handles range r of `iq` (every rank holds, or maps, what `needed_from` says).
    The only exchange: packet sizes (all_gather) and the packets themselves (padded gather) to rank 0, which has meanwhile put
    range 0 through its pipeline and then walks them.  Returns (messages, counters) on rank 0, None elsewhere.
    resident = (first sample held, device address): the rank's samples are already in HBM; then `iq` may be None, with
    `nsamples` = the capture's length and `histories` = (the 326 samples before the warm-up, the 326 before the range) as bytes.
    phases (a dict) collects this rank's wall time per phase, in ms, summed over calls.
    out: rank 0 builds every message straight into this mgpu_msg array (nothing is copied at the end)."""
    import time
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(), dist.get_rank()
    device = device or torch.device("cpu")
    if iq is not None:
        iq = np.ascontiguousarray(iq).view(np.uint8).reshape(-1)
    n = nsamples if nsamples is not None else iq.size // _FMT_BYTES[d.fmt]
    first, last = shard_ranges(n, world)[rank]
    t = [time.perf_counter()]

    def lap(name):
        t.append(time.perf_counter())
        if phases is not None:
            phases[name] = phases.get(name, 0.0) + (t[-1] - t[-2]) * 1e3

    if rank == 0:
        run_first_range(d, iq, last, resident, out)
        lap("first_range_pipeline")
        pk = np.zeros(0, dtype=np.uint8)
    else:
        pk = rank_packets(d, iq, first, last, resident, histories, phases, lap)
    gathered, sizes = None, [0] * world
    if world > 1:
        size = torch.tensor([pk.size], dtype=torch.int64, device=device)
        szs = [torch.zeros_like(size) for _ in range(world)]
        dist.all_gather(szs, size)
        sizes = [int(s.item()) for s in szs]
        buf = torch.zeros(max(max(sizes), 1), dtype=torch.uint8, device=device)
        if pk.size:
            buf[:pk.size] = torch.from_numpy(pk).to(device)
        gathered = [torch.zeros_like(buf) for _ in range(world)] if rank == 0 else None
        dist.gather(buf, gathered, dst=0)
        lap("gather_packets")
    if rank != 0:
        return None
    for r in range(1, world):
        if sizes[r]:
            d.walk_packets(gathered[r][:sizes[r]].cpu().numpy())
    lap("walk_packets")
    d.finish()
    res = d.collect(out=out) if out is not None else d.collect()
    lap("collect")
    return res
