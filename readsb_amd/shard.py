"""One capture sharded by buffer ranges over several GPUs (BASELINE config 5).

Buffers of 131072 samples are independent except for the ICAO filter, and the GPU pre-screen
needs the adder addresses of the whole capture.  Every rank therefore sweeps its range twice — once for its
adder bitmap, once (after the bitmaps have been OR-ed over all ranks: the exchange step, 2 MiB per rank)
through convert, sweep and pre-screen — and ships the surviving records (≈ 45 bytes per 1000 samples) to one
rank, which runs the ordered walk and builds the messages exactly as for an unsharded stream.  The result is
the unsharded message list, bit for bit.  What it costs: the second sweep (a production version keeps the
record pools of pass 1 in HBM instead — 2 bytes per sample, there is room), and the walk is not sharded (on
the benchmark stream it is ≈ 40 % of one GPU's kernel time per sample; `Resolver::parallel_walk` is the
algorithm that shards it, with ranks in place of threads).

`shard_ranges` / `run_shard_pass` / `walk_all` are the pieces; `demodulate_sharded_local` runs them for all
shards in one process (tests, single GPU); `demodulate_sharded` is the torch.distributed version (one rank =
one shard; `gloo` with host tensors or `nccl` with device tensors)."""
import numpy as np

from .binding import _FMT_BYTES

BUF = 131072
TRAILING = 326


def shard_ranges(nsamples, nshards, buf=BUF):
    """Contiguous whole-buffer ranges [first, last) in samples, one per shard (possibly empty at the end)."""
    nbuf = (nsamples + buf - 1) // buf
    out = []
    for s in range(nshards):
        b0, b1 = nbuf * s // nshards, nbuf * (s + 1) // nshards
        out.append((min(b0 * buf, nsamples), min(b1 * buf, nsamples)))
    return out


def _feed(d, iq, first, last, bps, mode):
    hist = None if first == 0 else iq[(first - TRAILING) * bps:first * bps]
    d.shard_begin(first, hist, mode)
    cap = int(d.cfg.max_samples)
    cap -= cap % BUF
    off = first
    while off < last:
        k = min(cap, last - off)
        d.feed_iq(iq[off * bps:(off + k) * bps])
        off += k


def run_shard_pass(d, iq, first, last, mode, global_bitmap=None):
    """mode 1 -> this shard's adder bitmap; mode 2 (needs the global bitmap) -> its record packets."""
    bps = _FMT_BYTES[d.fmt]
    d.reset()
    if mode == 2:
        d.set_adder_bitmap(global_bitmap)
    if last > first:
        _feed(d, iq, first, last, bps, mode)
    return d.adder_bitmap() if mode == 1 else d.shard_packets()


def run_shard_pass_resident(d, iq, first, last, mode, global_bitmap=None, resident=None, history=None):
    """As run_shard_pass, with the shard's IQ samples already in device memory (`resident` = (first sample held, device
    address of it)): the benchmark's form — nothing but the 326 history samples crosses PCIe inside the timed region
    (`history` = their bytes when the rank does not hold the capture in host memory)."""
    bps = _FMT_BYTES[d.fmt]
    d.reset()
    if mode == 2:
        d.set_adder_bitmap(global_bitmap)
    if last > first:
        hist = None if first == 0 else (history if history is not None else iq[(first - TRAILING) * bps:first * bps])
        d.shard_begin(first, hist, mode)
        base_first, base_ptr = resident
        cap = int(d.cfg.max_samples)
        cap -= cap % BUF
        off = first
        while off < last:
            k = min(cap, last - off)
            d.feed_resident(k, base_ptr + (off - base_first) * bps)
            off += k
    return d.adder_bitmap() if mode == 1 else d.shard_packets()


def walk_all(d, packets_in_stream_order):
    """The ordered walk over every shard's packets on one context: (messages, counters)."""
    d.reset()
    for pk in packets_in_stream_order:
        if pk.size:
            d.walk_packets(pk)
    d.finish()
    return d.collect()


def demodulate_sharded_local(d, iq, nshards):
    """All shards one after the other on one Demodulator (no communication): the algorithm's reference run."""
    iq = np.ascontiguousarray(iq).view(np.uint8).reshape(-1)
    n = iq.size // _FMT_BYTES[d.fmt]
    ranges = shard_ranges(n, nshards)
    bitmap = np.zeros(1 << 19, dtype=np.uint32)
    for first, last in ranges:
        bitmap |= run_shard_pass(d, iq, first, last, 1)
    packets = [run_shard_pass(d, iq, first, last, 2, bitmap).copy() for first, last in ranges]   # (copies: the next pass reuses the buffer)
    return walk_all(d, packets)


def demodulate_sharded(d, iq, device=None, dst=0, resident=None, nsamples=None, history=None, phases=None):
    """torch.distributed version: rank r handles range r of `iq` (every rank holds, or maps, the capture).
    Exchange 1: all_gather of the 2 MiB adder bitmaps, OR.  Exchange 2: packet sizes (all_gather) and the
    packets themselves (padded gather) to `dst`, which walks them.  Returns (messages, counters) on dst, None elsewhere.
    resident = (first sample, device address): the rank's range is already in HBM (run_shard_pass_resident); then `iq` may be
    None, with `nsamples` = the capture's length and `history` = the 326 samples before the rank's range.
    phases (a dict) collects this rank's wall time per phase, in ms, summed over calls."""
    import time
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(), dist.get_rank()
    device = device or torch.device("cpu")
    if iq is not None:
        iq = np.ascontiguousarray(iq).view(np.uint8).reshape(-1)
    n = nsamples if nsamples is not None else iq.size // _FMT_BYTES[d.fmt]
    first, last = shard_ranges(n, world)[rank]
    def shard_pass(mode, bitmap=None):
        if resident is not None:
            return run_shard_pass_resident(d, iq, first, last, mode, bitmap, resident, history)
        return run_shard_pass(d, iq, first, last, mode, bitmap)

    t = [time.perf_counter()]

    def lap(name):
        t.append(time.perf_counter())
        if phases is not None:
            phases[name] = phases.get(name, 0.0) + (t[-1] - t[-2]) * 1e3

    mine = torch.from_numpy(shard_pass(1).view(np.int32)).to(device)
    lap("pass1_adder_bitmap")
    allmaps = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(allmaps, mine)
    bitmap = allmaps[0]
    for m in allmaps[1:]:
        bitmap = torch.bitwise_or(bitmap, m)
    gb = bitmap.cpu().numpy().view(np.uint32)
    lap("bitmap_exchange")
    pk = shard_pass(2, gb)
    lap("pass2_packets")
    if world == 1:                      # nothing to gather: the packets are walked where the shard pass left them
        d.walk_own_packets()
        lap("walk_and_build")
        d.finish()
        res = d.collect()
        lap("collect")
        return res
    size = torch.tensor([pk.size], dtype=torch.int64, device=device)
    sizes = [torch.zeros_like(size) for _ in range(world)]
    dist.all_gather(sizes, size)
    sizes = [int(s.item()) for s in sizes]
    buf = torch.zeros(max(max(sizes), 1), dtype=torch.uint8, device=device)
    if pk.size:
        buf[:pk.size] = torch.from_numpy(pk).to(device)
    gathered = [torch.zeros_like(buf) for _ in range(world)] if rank == dst else None
    dist.gather(buf, gathered, dst=dst)
    if rank != dst:
        return None
    return walk_all(d, [gathered[r][:sizes[r]].cpu().numpy() for r in range(world)])
