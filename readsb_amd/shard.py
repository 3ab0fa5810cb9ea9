"""One capture sharded by buffer ranges over several GPUs (BASELINE config 5).

Buffers of 131072 samples are independent except for the ICAO filter: scoring reads it, accepted clean DF17 /
DF11-IID0 frames write it, and it expires on the stream's 60 s clock (icao_filter.c:96-130, readsb.c:1227-1231).

**The form that ships (round 4; `ShardStreamRank`, `run_stream_protocol`, `demodulate_sharded_stream*`): every rank walks and
builds its OWN range.**
  1. the expiry schedule of the whole capture is derived up front: a pre-pass over the ~6 % of buffers an expiry can follow gives
     every rank the clocks its share of the schedule needs (`expiry_windows`, `schedule_from_window_estimates`);
  2. every rank puts two filter generations of warm-up + its range through the ordinary pipeline with that schedule IMPOSED
     (`mgpu_shard_stream_begin / _mark / _end`): its messages and counters are final if the schedule and the filter state at its
     seam were right;
  3. rounds over (schedule, seam states) confirm or repair both (`protocol_round` = `mgpu_shard_round`, a fixed point: in practice
     one round);
  4. the ranges' messages are gathered on rank 0 as in config 4, the integer counters add up, and the two order-dependent double
     sums (demod_2400.c:445-447, 474-479) are re-added exactly from per-block partial sums (`prepare_sum_blocks`,
     `combine_ranges`, seqsum.cpp) — O(blocks) on the combining rank, 1.5 % of the unsharded time on the one-hour capture.
The result is the unsharded message list and every counter, bit for bit (tests/test_gpu_shard.py, tests/test_shard_walk.py;
the same protocol in C over RCCL: readsb_amd/host/readsb_gpu_shard.c).  With one rank this IS the unsharded pipeline.

The older forms: rounds 2-3 (the first part of this file: `run_shard_pass` / `rank_packets` / `demodulate_sharded[_local]`): ranks r > 0 sweep a 120 s warm-up for its adder addresses, then their range, and ship the surviving
records as packets; rank 0 walks every range's packets in order (`mgpu_walk_packets`).  Exact as well, but the walk does not shrink
with the number of ranks; and round 4's first cut (`ShardWalkRank`, `demodulate_sharded_walk*`: every rank walks its own range's
PACKETS).  Both kept because bench.py --config5-form packets and the tests compare the forms with each other.

`gloo` with host tensors or `nccl` (= RCCL) with device tensors for both."""
import numpy as np

from .binding import _FMT_BYTES, MSG_DTYPE

BUF = 131072
TRAILING = 326
SAMPLE_RATE = 2400000
FILTER_TTL_S = 60                       # MODES_ICAO_FILTER_TTL, readsb.h:315: one generation
# Two generations of samples in whole buffers.  A generation is not 60 s sharp: the expiry is tested after a buffer against the
# timestamp of the buffer's last scored candidate (anywhere within the buffer's 54.6 ms) and re-armed 60 s after THAT
# (readsb.c:1227-1231, demod_2400.c:412-414), so two consecutive expiries lie up to 60 s + two buffers apart: 2 x (60 s + 2 buffers),
# one buffer to spare.  (Rounds 2-3 took 120 s + 1 buffer: 0.11 s short of the bound.)
WARMUP = (2 * FILTER_TTL_S * SAMPLE_RATE + BUF - 1) // BUF * BUF + 5 * BUF


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


# ---------------------------------------------------------------------------------------------------------------------------
# Round 4: the ordered walk itself sharded.  Every rank walks and builds its OWN range; what crosses ranks is a few KB per round
# (every buffer's end clock, the filter state at every range's two ends) and, at the end, the messages — as in config 4.
# include/modes_gpu.h ("config 5 with the ordered walk itself sharded") states the protocol; DESIGN.md §5 why it is exact.
# ---------------------------------------------------------------------------------------------------------------------------

def _pack(clocks, state_first, state_end):
    head = np.array([clocks.size, len(state_first), len(state_end)], dtype=np.int64)
    return head.tobytes() + np.ascontiguousarray(clocks, dtype=np.int64).tobytes() + state_first + state_end


def _unpack(b):
    b = bytes(b)
    nc, n0, n1 = (int(x) for x in np.frombuffer(b[:24], dtype=np.int64))
    o = 24
    clocks = np.frombuffer(b[o:o + 8 * nc], dtype=np.int64)
    o += 8 * nc
    return clocks, b[o:o + n0], b[o + n0:o + n0 + n1]


class ShardWalkRank:
    """One rank of a capture of `nsamples` samples walked by `world` ranks (whole-buffer ranges, shard_ranges)."""

    def __init__(self, d, rank, world, nsamples, keep_packets=False, out=None):
        self.d, self.rank, self.world, self.n = d, rank, world, int(nsamples)
        self.out = out                         # a standing mgpu_msg array the range's messages are built into (none: the library's list, copied out)
        self.est = None
        self.first, self.last = shard_ranges(self.n, world)[rank]
        self.ws = warmup_start(self.first)
        self.nbuf = (self.last - self.first + BUF - 1) // BUF
        self.keep_packets = keep_packets       # one context plays several ranks (tests, bench --emulate-ranks): its packets are copied out
        self.packets = None
        self.import_state = None               # the state at `first`, from the rank before, once this rank's own warm-up proved wrong
        self.result = None                     # (clocks, state_first, state_end) of the last walk
        self.used = None                       # (schedule within the rank's window, imported state) of the last walk
        self.msgs = self.counters = self.noise = None
        self.walks = 0
        self.ms = {}

    def _lap(self, name, t0):
        import time
        self.ms[name] = self.ms.get(name, 0.0) + (time.perf_counter() - t0) * 1e3

    def gpu_phase(self, iq=None, resident=None, histories=None):
        """Warm-up + range through convert / sweep / slice / pre-screen; the live records stay on the host as packets.
        iq: the capture (host bytes), or resident = (first sample held, device address) with histories = {sample: the 326
        IQ samples before it, as bytes} for the warm-up's first sample."""
        import time
        t0 = time.perf_counter()
        d = self.d
        if self.last <= self.first:
            self.packets = np.zeros(0, dtype=np.uint8)
            return
        bps = _FMT_BYTES[d.fmt]
        d.reset()
        hist = None
        if self.ws > 0:
            hist = histories[self.ws] if histories is not None else iq[(self.ws - TRAILING) * bps:self.ws * bps]
        d.shard_begin(self.ws, hist, 2)
        for a, b in ((self.ws, self.first), (self.first, self.last)):     # two feeds: no packet straddles the range's first sample
            if b > a:
                if resident is not None:
                    _feed_resident(d, a, b, bps, resident)
                else:
                    _feed_host(d, iq, a, b, bps)
        self._lap("gpu_phase", t0)
        t0 = time.perf_counter()
        self.est = d.shard_clock_estimate(self.first, self.nbuf).copy()     # (the fetcher estimated every chunk as it came)
        self._lap("clock_estimate", t0)
        if self.keep_packets:
            self.packets = d.shard_packets().copy()

    def estimate(self):
        return self.est if self.est is not None else np.zeros(0, dtype=np.int64)

    def walk(self, sched_ts):
        """Walk (unless nothing this rank depends on has changed since its last walk) and collect the range's messages."""
        import time
        if self.nbuf == 0:
            self.result = (np.zeros(0, dtype=np.int64), b"", b"")
            self.msgs, self.counters, self.noise = np.zeros(0, dtype=MSG_DTYPE), None, np.zeros(0)
            return self.result
        lo, hi = self.ws * 5, self.last * 5
        mine = sched_ts[(sched_ts >= lo) & (sched_ts < hi)]
        nbefore = int((sched_ts < self.first * 5).sum())         # (a cold start numbers its expiries from the schedule)
        key = (mine.tobytes(), nbefore, self.import_state)
        if self.used == key:
            return self.result
        t0 = time.perf_counter()
        d, out = self.d, self.out
        d.set_message_buffer(out)
        self.result = d.shard_walk(self.first, self.nbuf, sched_ts, self.import_state, self.packets)
        self.used = key
        self.walks += 1
        tm = d.timing()
        self.ms["walk_warmup"] = self.ms.get("walk_warmup", 0.0) + tm["d2h_ms"]
        self.ms["walk_range"] = self.ms.get("walk_range", 0.0) + tm["resolve_ms"]
        self.ms["build_range"] = self.ms.get("build_range", 0.0) + tm["build_ms"]
        self._lap("walk_call", t0)
        t0 = time.perf_counter()
        self.msgs, self.counters = d.collect(out=out) if out is not None else d.collect()
        self.noise = d.shard_noise_terms()
        self._lap("collect", t0)
        return self.result


def eof_clock(nsamples, startup_ms):
    """The zero-length buffer ifileRun pushes after a capture that is a whole number of buffers (sdr_ifile.c:223-237): its clock."""
    return (nsamples * 5) // 12000 + startup_ms if nsamples % BUF == 0 else None


def schedule_from_clocks(per_rank_clocks, nsamples, startup_ms, filter_clock=0):
    """All buffers' end clocks (rank order = stream order) -> the sampleTimestamps of the buffers the filter expires after."""
    from .binding import flip_schedule
    clocks = np.concatenate([np.asarray(c, dtype=np.int64) for c in per_rank_clocks] + [np.zeros(0, dtype=np.int64)])
    e = eof_clock(nsamples, startup_ms)
    if e is not None:
        clocks = np.concatenate([clocks, np.array([e], dtype=np.int64)])
    return flip_schedule(clocks, startup_ms, filter_clock).astype(np.int64) * (BUF * 5)


def protocol_round(sched_ts, gathered, nsamples, startup_ms, filter_clock=0):
    """What every rank concludes from a round's all-gather (the same on every rank: no further exchange needed) — the library's
    mgpu_shard_round, which a C host calls too.  gathered[r] = (clocks, state_first, state_end) of rank r.
    -> (done, next schedule, {rank: state to import})."""
    from .binding import shard_round
    done, nxt, imp = shard_round(sched_ts, gathered, nsamples, startup_ms, filter_clock)
    return done, nxt, {r: gathered[src][2] for r, src in imp.items()}


_INT_FIELDS = ["demod_preambles", "demod_rejected_bad", "demod_rejected_unknown_icao", "demod_accepted", "demod_preamblePhase",
               "demod_bestPhase", "strong_signal_count", "signal_power_count", "noise_power_count", "samples_processed",
               "samples_lost", "nbuffers", "demod_modeac"]


def prepare_sum_blocks(msgs, earlier_counters, sig_terms=None):
    """What a rank adds to its range's results so that the combining rank need not re-add every message's signal power one by
    one: the block-wise form of that sequential sum (readsb_amd/csrc/seqsum.cpp), predicted from the earlier ranges' own totals.
    sig_terms (round 6): the 8-byte numerators of the messages' signal powers as the builder logged them during the pass
    (Demodulator.shard_signal_terms) — the same blocks from an eighth of the memory."""
    from .binding import seqsum_blocks, seqsum_blocks_terms
    approx = float(sum(c["signal_power_sum"] for c in earlier_counters if c is not None))
    if sig_terms is not None and len(sig_terms) == len(msgs) and len(msgs):
        return seqsum_blocks_terms(approx, sig_terms)
    return seqsum_blocks(approx, np.ascontiguousarray(msgs))


def combine_ranges(parts, nsamples, nflips, stats=None, concat=True):
    """parts[r] = (messages, counters, noise terms[, sum blocks]) of rank r, in rank order -> the unsharded run's (messages,
    counters).  Integer counters add up; the two double sums are the reference's SEQUENTIAL sums (demod_2400.c:445-447, 474-479),
    re-added in stream order: the per-buffer noise terms one by one (one per buffer), the messages' signal powers block-wise
    (prepare_sum_blocks) or, without blocks, message by message.  concat=False: the messages stay the list of per-range arrays
    they arrived as (an aggregator's gather lands them in one buffer anyway)."""
    from .binding import seqsum, seqsum_signal_power, seqsum_apply
    total = None
    sig = noise = 0.0
    peak = 0.0
    for part in parts:
        msgs, cnt, terms = part[:3]
        blocks = part[3] if len(part) > 3 else None
        if cnt is None:
            continue
        if total is None:
            total = {k: (list(v) if isinstance(v, list) else v) for k, v in cnt.items()}
            for k in _INT_FIELDS:
                total[k] = [0] * len(cnt[k]) if isinstance(cnt[k], list) else 0
        for k in _INT_FIELDS:
            if isinstance(cnt[k], list):
                total[k] = [a + b for a, b in zip(total[k], cnt[k])]
            else:
                total[k] += cnt[k]
        if blocks is not None:
            sig, fb = seqsum_apply(sig, np.ascontiguousarray(msgs), blocks)
            if stats is not None:
                stats["sum_blocks"] = stats.get("sum_blocks", 0) + len(blocks)
                stats["sum_blocks_readded"] = stats.get("sum_blocks_readded", 0) + fb
        else:
            sig = seqsum_signal_power(sig, np.ascontiguousarray(msgs))
        noise = seqsum(noise, terms)
        peak = max(peak, cnt["peak_signal_power"])
    if nsamples % BUF == 0:                         # the EOF buffer: 0 / 0 in the converter (convert.c:101-107), mgpu_finish
        noise += float("nan")
        total["samples_lost"] += BUF
        total["nbuffers"] += 1
    total["nflips"] = int(nflips)
    total["signal_power_sum"], total["noise_power_sum"], total["peak_signal_power"] = sig, noise, peak
    if not concat:
        return [p[0] for p in parts], total
    msgs = np.concatenate([p[0] for p in parts]) if len(parts) > 1 else parts[0][0]
    return msgs, total


def run_walk_protocol(ranks, exchange, nsamples, startup_ms, filter_clock=0, max_rounds=None, stats=None):
    """The rounds of the protocol for the ranks this process plays.  exchange(list of bytes, one per own rank) -> list of bytes
    of ALL ranks in rank order (an all-gather).  -> the final schedule (sampleTimestamps)."""
    est = exchange([_pack(r.estimate(), b"", b"") for r in ranks])
    sched = schedule_from_clocks([_unpack(b)[0] for b in est], nsamples, startup_ms, filter_clock)
    world = len(est)
    rounds = 0
    while True:
        rounds += 1
        if max_rounds is None:
            max_rounds = world + 72
        if rounds > max_rounds:
            raise RuntimeError("sharded walk: the schedule / seam iteration did not settle")
        got = [_unpack(b) for b in exchange([_pack(*r.walk(sched)) for r in ranks])]
        done, nxt, imports = protocol_round(sched, got, nsamples, startup_ms, filter_clock)
        if stats is not None:
            stats.setdefault("rounds", 0)
            stats["rounds"] = rounds
            stats["seam_failures"] = stats.get("seam_failures", 0) + len(imports)
            stats["schedule_changes"] = stats.get("schedule_changes", 0) + (0 if (nxt.size == sched.size and (nxt == sched).all()) else 1)
        if done:
            return sched
        sched = nxt
        for r in ranks:
            if r.rank in imports:
                r.import_state = imports[r.rank]


def demodulate_sharded_walk_local(d, iq, nshards, stats=None):
    """All ranks of the sharded walk played by ONE Demodulator, one after the other (tests, single GPU): the same protocol, the
    all-gather replaced by a list."""
    iq = np.ascontiguousarray(iq).view(np.uint8).reshape(-1)
    n = iq.size // _FMT_BYTES[d.fmt]
    ranks = [ShardWalkRank(d, r, nshards, n, keep_packets=True) for r in range(nshards)]
    for r in ranks:
        r.gpu_phase(iq)
    startup = int(d.cfg.startup_time_ms)
    fc = int(d.cfg.filter_clock)
    sched = run_walk_protocol(ranks, lambda payloads: payloads, n, startup, fc, stats=stats)
    if stats is not None:
        stats["walks"] = [r.walks for r in ranks]
        stats["imported"] = [r.import_state is not None for r in ranks]
    parts = [(r.msgs, r.counters, r.noise, prepare_sum_blocks(r.msgs, [q.counters for q in ranks[:r.rank]])) for r in ranks]
    return combine_ranges(parts, n, len(sched) + (1 if fc == 1 else 0), stats)


def _all_gather_bytes(payload, device):
    import torch
    import torch.distributed as dist
    world = dist.get_world_size()
    size = torch.tensor([len(payload)], dtype=torch.int64, device=device)
    sizes = [torch.zeros_like(size) for _ in range(world)]
    dist.all_gather(sizes, size)
    sizes = [int(x.item()) for x in sizes]
    buf = torch.zeros(max(max(sizes), 1), dtype=torch.uint8, device=device)
    if payload:
        buf[:len(payload)] = torch.frombuffer(bytearray(payload), dtype=torch.uint8).to(device)
    parts = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(parts, buf)
    return [bytes(parts[r][:sizes[r]].cpu().numpy().tobytes()) for r in range(world)]


def demodulate_sharded_walk(d, iq, device=None, resident=None, nsamples=None, histories=None, phases=None, out=None, stats=None):
    """torch.distributed version: this process is ONE rank.  Collectives: an all-gather of a few KB per protocol round (clocks and
    states), then the gather of every range's messages, counters and noise terms to rank 0.  Returns (messages, counters) on rank
    0, None elsewhere."""
    import time
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(), dist.get_rank()
    device = device or torch.device("cpu")
    if iq is not None:
        iq = np.ascontiguousarray(iq).view(np.uint8).reshape(-1)
    n = nsamples if nsamples is not None else iq.size // _FMT_BYTES[d.fmt]
    me = ShardWalkRank(d, rank, world, n)
    me.gpu_phase(iq, resident, histories)
    startup, fc = int(d.cfg.startup_time_ms), int(d.cfg.filter_clock)
    t0 = time.perf_counter()
    sched = run_walk_protocol([me], lambda payloads: _all_gather_bytes(payloads[0], device), n, startup, fc, stats=stats)
    return _gather_and_combine(me, sched, n, fc, device, phases, stats, t0)


def _gather_and_combine(me, sched, n, fc, device, phases, stats, t0, concat=True):
    """Every range's messages, counters, noise terms and sum blocks to rank 0 (an all-gather of the small parts, a gather of the
    messages), combined there.  -> (messages, counters) on rank 0, None elsewhere."""
    import time
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(), dist.get_rank()
    t1 = time.perf_counter()
    # ---- the ranges' results to rank 0 ----
    # wire form of a rank's small part: [messages: i64][struct mgpu_counters][noise terms: f64 ...] — fixed layouts, as in the C host
    # (readsb_gpu_shard.c); a rank without a range sends the count alone
    import ctypes as C
    from .binding import Counters
    ksz = C.sizeof(Counters)

    def unpack_meta(m):
        if len(m) <= 8:
            return None, np.zeros(0)
        return Counters.from_buffer_copy(m[8:8 + ksz]).as_dict(), np.frombuffer(m[8 + ksz:], dtype=np.float64)
    cnt = me.counters
    blob = b""
    if cnt is not None:
        blob = bytes(Counters.from_dict(cnt)) + np.ascontiguousarray(me.noise, dtype=np.float64).tobytes()
    metas = _all_gather_bytes(np.array([me.msgs.size], dtype=np.int64).tobytes() + blob, device)
    counts = [int(np.frombuffer(m[:8], dtype=np.int64)[0]) for m in metas]
    earlier = [unpack_meta(m)[0] for m in metas[:rank]]
    from .binding import SUM_BLOCK, SUM_BLOCK_DTYPE
    blocks = prepare_sum_blocks(me.msgs, earlier, getattr(me, "sig_terms", None))   # (every rank at once: its part of the sequential signal-power sum)
    rec, brec = MSG_DTYPE.itemsize, SUM_BLOCK_DTYPE.itemsize
    nblk = [(c + SUM_BLOCK - 1) // SUM_BLOCK for c in counts]
    buf = torch.zeros(max(max(c * rec + b * brec for c, b in zip(counts, nblk)), 1), dtype=torch.uint8, device=device)
    if me.msgs.size:
        mine = np.concatenate([np.ascontiguousarray(me.msgs).view(np.uint8).reshape(-1), blocks.view(np.uint8).reshape(-1)])
        buf[:mine.size] = torch.from_numpy(mine).to(device)
    gathered = [torch.zeros_like(buf) for _ in range(world)] if rank == 0 else None
    dist.gather(buf, gathered, dst=0)
    t2 = time.perf_counter()
    if phases is not None:
        for k, v in me.ms.items():
            phases[k] = phases.get(k, 0.0) + v
        phases["protocol_rounds_wall"] = phases.get("protocol_rounds_wall", 0.0) + (t1 - t0) * 1e3
        phases["gather_results"] = phases.get("gather_results", 0.0) + (t2 - t1) * 1e3
    if rank != 0:
        return None
    parts = []
    for r in range(world):
        raw = gathered[r][:counts[r] * rec + nblk[r] * brec]
        raw = raw.numpy() if raw.device.type == "cpu" else raw.cpu().numpy()      # (gloo: a view; RCCL: the D2H copy a host-side consumer needs anyway)
        msgs = raw[:counts[r] * rec].view(MSG_DTYPE)
        if len(metas[r]) > 8:
            c, terms = unpack_meta(metas[r])
            parts.append((msgs, c, terms, raw[counts[r] * rec:].view(SUM_BLOCK_DTYPE)))
        else:
            parts.append((msgs, None, np.zeros(0)))
    res = combine_ranges(parts, n, len(sched) + (1 if fc == 1 else 0), stats, concat=concat)
    if phases is not None:
        phases["combine"] = phases.get("combine", 0.0) + (time.perf_counter() - t2) * 1e3
    return res




# ---------------------------------------------------------------------------------------------------------------------------
# ... and the form in which a rank's walk and build OVERLAP its kernels: the schedule is derived before the pass, from a pre-pass
# over the few buffers an expiry can follow, and warm-up + range then go through the ordinary pipeline (mgpu_shard_stream_*).
# ---------------------------------------------------------------------------------------------------------------------------

def buffer_sys_ms(b, startup_ms):
    """sysTimestamp of buffer b (sdr_ifile.c:216): where its clock starts; its end clock lies within the 55 ms after."""
    return (np.asarray(b, dtype=np.int64) * (BUF * 5)) // 12000 + int(startup_ms)


def expiry_windows(nbuf_total, startup_ms, filter_clock=0):
    """The buffers an expiry of the ICAO filter CAN follow, as a boolean mask (mgpu_expiry_windows).  The k-th expiry follows the
    first buffer whose end clock reaches T_k, and T_(k+1) = (that buffer's end clock) + 60 s (readsb.c:1227-1231) with the end clock
    anywhere in the buffer's own 55 ms: T_k + 60 000 <= T_(k+1) < T_k + 60 111 (the buffer that reaches T_k starts within 55.6 ms of
    it).  So T_k lies in a window that widens by 111 ms per expiry, and only buffers whose 55 ms touch a window matter to the
    schedule: ~2 k of them for the k-th expiry."""
    from .binding import expiry_windows as _w
    return _w(nbuf_total, startup_ms, filter_clock, BUF)


class _Source:
    """Where a rank's samples are: host bytes, or resident in HBM (first sample held, device address, a function that gathers
    sample ranges into one device buffer)."""

    def __init__(self, d, iq=None, resident=None, histories=None, gather=None):
        self.d, self.iq, self.resident, self.histories, self.gather_fn = d, iq, resident, histories, gather
        self.bps = _FMT_BYTES[d.fmt]

    def history(self, sample):
        if sample == 0:
            return None
        if self.iq is not None:
            return self.iq[(sample - TRAILING) * self.bps:sample * self.bps]
        return self.histories[sample]

    def feed(self, a, b):
        if b <= a:
            return
        if self.resident is not None:
            _feed_resident(self.d, a, b, self.bps, self.resident)
        else:
            _feed_host(self.d, self.iq, a, b, self.bps)

    def feed_gathered(self, ranges):
        """The sample ranges as ONE stream (for the pre-pass: whole buffers, their order kept)."""
        total = sum(b - a for a, b in ranges)
        if self.iq is not None:
            fake = np.concatenate([self.iq[a * self.bps:b * self.bps] for a, b in ranges])
            _feed_host(self.d, fake, 0, total, self.bps)
        else:
            keep, ptr = self.gather_fn(ranges)
            _feed_resident(self.d, 0, total, self.bps, (0, ptr))
            del keep


class ShardStreamRank(ShardWalkRank):
    """A rank whose pass goes through the ordinary pipeline (mgpu_shard_stream_*): `prepass` estimates the end clocks of its buffers
    inside the expiry windows, `stream_pass` runs warm-up + range with the schedule imposed."""

    def __init__(self, d, rank, world, nsamples, source, out=None):
        super().__init__(d, rank, world, nsamples, keep_packets=False, out=out)
        self.src = source

    def prepass(self, startup_ms, filter_clock=0):
        """-> (global indices of this rank's buffers inside the expiry windows, their estimated end clocks)."""
        import time
        t0 = time.perf_counter()
        nb_total = (self.n + BUF - 1) // BUF
        mask = expiry_windows(nb_total, startup_ms, filter_clock)
        b0, b1 = self.first // BUF, (self.last + BUF - 1) // BUF
        idx = np.flatnonzero(mask[b0:b1]) + b0
        if idx.size and idx[-1] == nb_total - 1 and self.n % BUF:
            idx = idx[:-1]                                       # (a short last buffer does not fit the gathered stream's grid: its clock is guessed)
        if idx.size == 0:
            return idx, np.zeros(0, dtype=np.int64)
        runs, start = [], idx[0]
        for k in range(1, idx.size + 1):
            if k == idx.size or idx[k] != idx[k - 1] + 1:
                runs.append((int(start) * BUF, (int(idx[k - 1]) + 1) * BUF))
                if k < idx.size:
                    start = idx[k]
        d = self.d
        d.reset()
        d.shard_begin(0, None, 3)                                # clock estimates only: no packets, no per-record signal powers or windows
        self.src.feed_gathered(runs)
        est = d.shard_clock_estimate(0, idx.size).copy()
        # the gathered stream's buffer i stands for buffer idx[i]: what counts is how far into its own 55 ms the clock ends
        clocks = buffer_sys_ms(idx, startup_ms) + (est - buffer_sys_ms(np.arange(idx.size), startup_ms))
        self._lap("prepass", t0)
        self.ms["prepass_buffers"] = int(idx.size)
        return idx, clocks

    def stream_pass(self, sched_ts):
        import time
        if self.nbuf == 0:
            self.result = (np.zeros(0, dtype=np.int64), b"", b"")
            self.msgs, self.counters, self.noise = np.zeros(0, dtype=MSG_DTYPE), None, np.zeros(0)
            return self.result
        lo, hi = self.ws * 5, self.last * 5
        mine = sched_ts[(sched_ts >= lo) & (sched_ts < hi)]
        nbefore = int((sched_ts < self.first * 5).sum())
        key = (mine.tobytes(), nbefore, self.import_state)
        if self.used == key:
            return self.result
        t0 = time.perf_counter()
        d = self.d
        start = self.first if self.import_state is not None else self.ws
        cap = int(d.cfg.max_samples)
        cap -= cap % BUF
        # Warm-up and range as DEFERRED feeds when each is one feed call (round 5): the range's kernels run while the warm-up is
        # still being walked, the walker marks the range's begin itself (mgpu_shard_stream_mark then waits for nothing) — one
        # pipeline fill and drain per pass instead of two.  Otherwise the synchronous form.
        deferred = self.out is not None and (self.first - start) <= cap and (self.last - self.first) <= cap
        if deferred:
            d.set_deferred(True)
        try:
            d.shard_stream_begin(start, self.src.history(start), self.first, sched_ts, self.import_state)
            self.src.feed(start, self.first)                     # the warm-up (nothing with an imported state)
            d.shard_stream_mark()
            d.set_message_buffer(self.out)
            self.src.feed(self.first, self.last)
            self.result = d.shard_stream_end(self.nbuf)
            self.used = key
            self.walks += 1
            self._lap("stream_pass", t0)
            t0 = time.perf_counter()
            if deferred:
                if self.first > start:
                    d.collect_feed(self.out[:0])                 # the warm-up's feed: no messages
                self.msgs, self.counters = d.collect_feed(self.out, want_counters=True)
            else:
                self.msgs, self.counters = d.collect(out=self.out) if self.out is not None else d.collect()
            self.noise = d.shard_noise_terms()
            # 8 bytes per message for the sum blocks: a view of the library's array, valid until the context's next pass — a rank
            # with a context of its own uses it in place; one context playing several ranks keeps a copy (terms_view = False)
            self.sig_terms = d.shard_signal_terms() if getattr(self, "terms_view", False) else d.shard_signal_terms().copy()
        finally:
            if deferred:
                d.set_message_buffer(None)
                d.set_deferred(False)
        self._lap("collect", t0)
        return self.result


def schedule_from_window_estimates(parts, nsamples, startup_ms, filter_clock=0):
    """parts = every rank's (buffer indices, estimated end clocks) inside the expiry windows -> the schedule.  Buffers outside the
    windows cannot be followed by an expiry: their clocks are filled in with their start (anything within their 55 ms would do)."""
    nb_total = (nsamples + BUF - 1) // BUF
    clocks = buffer_sys_ms(np.arange(nb_total), startup_ms)
    for idx, c in parts:
        if len(idx):
            clocks[np.asarray(idx, dtype=np.int64)] = c
    return schedule_from_clocks([clocks], nsamples, startup_ms, filter_clock)


def run_stream_protocol(ranks, exchange, nsamples, startup_ms, filter_clock=0, stats=None):
    """The rounds for ShardStreamRanks.  exchange as in run_walk_protocol."""
    pre = [r.prepass(startup_ms, filter_clock) for r in ranks]
    got = exchange([np.array([len(i)], dtype=np.int64).tobytes() + np.asarray(i, dtype=np.int64).tobytes() + np.asarray(c, dtype=np.int64).tobytes() for i, c in pre])
    parts = []
    for b in got:
        b = bytes(b)
        k = int(np.frombuffer(b[:8], dtype=np.int64)[0])
        parts.append((np.frombuffer(b[8:8 + 8 * k], dtype=np.int64), np.frombuffer(b[8 + 8 * k:8 + 16 * k], dtype=np.int64)))
    sched = schedule_from_window_estimates(parts, nsamples, startup_ms, filter_clock)
    world = len(got)
    rounds = 0
    while True:
        rounds += 1
        if rounds > world + 72:
            raise RuntimeError("sharded walk: the schedule / seam iteration did not settle")
        res = [_unpack(b) for b in exchange([_pack(*r.stream_pass(sched)) for r in ranks])]
        done, nxt, imports = protocol_round(sched, res, nsamples, startup_ms, filter_clock)
        if stats is not None:
            stats["rounds"] = rounds
            stats["seam_failures"] = stats.get("seam_failures", 0) + len(imports)
            stats["schedule_changes"] = stats.get("schedule_changes", 0) + (0 if (nxt.size == sched.size and (nxt == sched).all()) else 1)
            stats["prepass_buffers"] = int(sum(len(p[0]) for p in parts))
        if done:
            return sched
        sched = nxt
        for r in ranks:
            if r.rank in imports:
                r.import_state = imports[r.rank]


def demodulate_sharded_stream_local(d, iq, nshards, stats=None, out_capacity=None):
    """All ranks of the stream form played by ONE Demodulator, one after the other (tests, single GPU).  out_capacity: every rank
    builds its messages into an array of its own of that many records (what a rank of demodulate_sharded_stream does with `out`) —
    and, where warm-up and range are one feed call each, passes through the pipeline with deferred feeds."""
    iq = np.ascontiguousarray(iq).view(np.uint8).reshape(-1)
    n = iq.size // _FMT_BYTES[d.fmt]
    src = _Source(d, iq=iq)
    ranks = [ShardStreamRank(d, r, nshards, n, src, out=None if out_capacity is None else np.empty(int(out_capacity), dtype=MSG_DTYPE)) for r in range(nshards)]
    startup, fc = int(d.cfg.startup_time_ms), int(d.cfg.filter_clock)
    sched = run_stream_protocol(ranks, lambda payloads: payloads, n, startup, fc, stats=stats)
    if stats is not None:
        stats["walks"] = [r.walks for r in ranks]
        stats["imported"] = [r.import_state is not None for r in ranks]
    parts = [(r.msgs, r.counters, r.noise, prepare_sum_blocks(r.msgs, [q.counters for q in ranks[:r.rank]], getattr(r, "sig_terms", None))) for r in ranks]
    return combine_ranges(parts, n, len(sched) + (1 if fc == 1 else 0), stats)


def demodulate_sharded_stream(d, iq=None, device=None, resident=None, nsamples=None, histories=None, gather=None, phases=None, out=None, stats=None,
                              concat=True):
    """torch.distributed version of the stream form: this process is ONE rank.  iq = the capture as host bytes, or resident = (first
    sample held, device address) with histories = {sample: the 326 IQ samples before it} for the samples this rank starts passes at
    (its warm-up's first, its range's first) and gather(ranges) -> (keep-alive, device address of those sample ranges made contiguous).
    Collectives: an all-gather of the pre-pass's clocks (a few KB), one of clocks + states per round, the gather of the results.
    concat=False: rank 0 gets the messages as the list of per-range arrays they arrived as (no 0.5 GB copy for the one-hour capture)."""
    import time
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(), dist.get_rank()
    device = device or torch.device("cpu")
    if iq is not None:
        iq = np.ascontiguousarray(iq).view(np.uint8).reshape(-1)
    n = nsamples if nsamples is not None else iq.size // _FMT_BYTES[d.fmt]
    src = _Source(d, iq=iq, resident=resident, histories=histories, gather=gather)
    me = ShardStreamRank(d, rank, world, n, src, out=out)
    startup, fc = int(d.cfg.startup_time_ms), int(d.cfg.filter_clock)
    t0 = time.perf_counter()
    sched = run_stream_protocol([me], lambda payloads: _all_gather_bytes(payloads[0], device), n, startup, fc, stats=stats)
    return _gather_and_combine(me, sched, n, fc, device, phases, stats, t0, concat)
