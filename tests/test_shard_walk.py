"""CPU: BASELINE config 5 with the ordered walk itself sharded over the ranks (include/modes_gpu.h "config 5 with the ordered
walk itself sharded", readsb_amd/shard.py) — the parts that need no GPU:
  * the protocol on synthetic record streams (mgpu_selftest_shard_walk: N ranks, warm-ups from an empty filter, imposed expiry
    schedule, rounds over schedule + seam states) against the serial walk: decisions, end clocks, counts, final filter state;
  * the expiry schedule from end clocks (mgpu_flip_schedule) against a restatement of readsb.c:1227-1231;
  * what a round concludes from its all-gather (protocol_round), how ranges are combined (combine_ranges: sequential double
    sums re-added in stream order), and the rounds' control flow over a real world-size-2 gloo all-gather with scripted ranks.
The runs on real captures are in tests/test_gpu_shard.py."""
import ctypes as C
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

import helpers


def _selftest():
    import readsb_amd
    lib = readsb_amd.load_library()
    f = lib.mgpu_selftest_shard_walk
    f.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64)]
    f.restype = C.c_int
    return f


# (seed, chunks, buffers per chunk, aircraft, front-loaded aircraft, ranks, ranges per chunk, flags) -> what the run must exercise
CASES = [
    ((1, 40, 128, 200, 0, 4, 4, 0), "plain"),
    ((2, 40, 128, 200, 0, 8, 1, 0), "serial chunk walks"),
    ((4, 30, 512, 1000, 0, 3, 8, 0), "seam"),                       # a seam fails, the rank behind it imports
    ((1486393352, 57, 128, 3000, 3000, 8, 4, 0), "seam"),           # table-size hysteresis: several seams, several rounds
    ((68494888361, 14, 256, 3000, 0, 3, 8, 3), "schedule"),         # a wrong first schedule is corrected by the rounds
    ((872945345143, 50, 256, 1000, 8000, 8, 4, 2), "both"),
    ((431205687120, 572, 32, 20, 0, 5, 1, 1), "plain"),             # 17 expiries, small chunks, crude first clocks
]


@pytest.mark.parametrize("args,kind", CASES, ids=[f"{c[1]}-{c[0][0]}" for c in CASES])
def test_protocol_equals_serial_walk(built, args, kind):
    st = (C.c_uint64 * 6)()
    rc = _selftest()(*args, st)
    rounds, walks, seam_failures, sched_changes, expiries, imported = list(st)
    assert rc == 0, (rc, list(st))
    assert walks >= args[5] and expiries >= 3
    if kind in ("seam", "both"):
        assert seam_failures >= 1 and imported >= 1 and rounds >= 2
    if kind in ("schedule", "both"):
        assert sched_changes >= 1 and rounds >= 2
    if kind == "plain":
        assert rounds == 1 and walks == args[5]                      # estimated clocks right, every warm-up right: one walk per rank


def _schedule_restated(clocks, startup, mode):
    next_flip = startup + 60000 if mode == 1 else 0                 # static next_flip = 0 (readsb.c:1227); mode 1: one expiry before buffer 0
    out = []
    for b, c in enumerate(clocks):
        if c >= next_flip:                                          # readsb.c:1228-1231
            out.append(b)
            next_flip = c + 60000
    return out


@pytest.mark.parametrize("mode", [0, 1])
def test_flip_schedule_is_the_reference_rule(built, mode):
    from readsb_amd.binding import flip_schedule
    rng = np.random.default_rng(5 + mode)
    startup = 1_700_000_000_000
    nbuf = 9000                                                     # 8 minutes of buffers
    starts = startup + (np.arange(nbuf) * 131072 * 5) // 12000
    clocks = starts + rng.integers(0, 55, size=nbuf)                # the last scored candidate: anywhere in the buffer
    clocks[rng.integers(0, nbuf, size=300)] = starts[0]             # (stale clocks are legal input)
    got = flip_schedule(clocks, startup, mode)
    assert list(got) == _schedule_restated(clocks, startup, mode)
    assert len(got) >= 8
    assert list(flip_schedule(np.zeros(0, dtype=np.int64), startup, mode)) == []


@pytest.mark.parametrize("mode", [0, 1])
def test_expiry_windows_hold_every_expiry(built, mode):
    """The stream form's pre-pass looks only at the buffers inside shard.expiry_windows: every expiry of every legal clock
    assignment (each buffer's end clock anywhere in its own 55 ms) must follow one of them; and they are a small part of a capture."""
    from readsb_amd import shard
    from readsb_amd.binding import flip_schedule
    rng = np.random.default_rng(11 + mode)
    startup = 1_700_000_123_456
    for nb in (1, 5, 1098, 1099, 1100, 1101, 2500, 66000):
        mask = shard.expiry_windows(nb, startup, mode)
        s = shard.buffer_sys_ms(np.arange(nb), startup)
        for trial in range(40 if nb < 60000 else 6):
            kind = trial % 4
            off = rng.integers(0, 55, size=nb) if kind == 0 else np.full(nb, 54) if kind == 1 else np.zeros(nb, dtype=np.int64) if kind == 2 else rng.choice([0, 54], size=nb)
            f = flip_schedule(s + off, startup, mode).astype(np.int64)
            assert mask[f].all(), (nb, kind, f[~mask[f]][:5])
    assert shard.expiry_windows(66000, startup, mode).mean() < 0.07          # the one-hour capture: ~6 % of its buffers
    # the schedule from window estimates alone = the schedule from all clocks
    nb = 30000
    s = shard.buffer_sys_ms(np.arange(nb), startup)
    clocks = s + rng.integers(0, 55, size=nb)
    mask = shard.expiry_windows(nb, startup, mode)
    idx = np.flatnonzero(mask)
    halves = [(idx[: idx.size // 3], clocks[idx[: idx.size // 3]]), (idx[idx.size // 3:], clocks[idx[idx.size // 3:]])]
    n = nb * 131072 - 5
    got = shard.schedule_from_window_estimates(halves, n, startup, mode)
    want = shard.schedule_from_clocks([clocks], n, startup, mode)
    assert list(got) == list(want) and len(got) >= 25


def test_round_conclusions_and_combination(built):
    from readsb_amd import shard
    n = 131072 * 6
    startup = 1000
    clocks = [np.array([1010, 1070], dtype=np.int64), np.zeros(0, dtype=np.int64), np.array([1120, 1180], dtype=np.int64), np.array([1230, 61075], dtype=np.int64)]
    sched = shard.schedule_from_clocks(clocks, n, startup)
    assert list(sched) == [0, 5 * 131072 * 5]                       # after buffer 0 (next_flip = 0) and after buffer 5 (61075 >= 1010 + 60000)
    ok = [(clocks[0], b"", b"A"), (clocks[1], b"", b""), (clocks[2], b"A", b"B"), (clocks[3], b"B", b"C")]
    done, nxt, imports = shard.protocol_round(sched, ok, n, startup)
    assert done and not imports and list(nxt) == list(sched)        # the empty range passes its neighbour's state through
    bad = [ok[0], ok[1], (clocks[2], b"X", b"B"), ok[3]]
    done, nxt, imports = shard.protocol_round(sched, bad, n, startup)
    assert not done and imports == {2: b"A"}
    done, nxt, imports = shard.protocol_round(sched[:1], ok, n, startup)
    assert not done and not imports and list(nxt) == list(sched)    # a schedule the clocks do not reproduce
    # a whole number of buffers: the EOF buffer's clock is part of the chain
    assert shard.eof_clock(n, startup) == (n * 5) // 12000 + startup and shard.eof_clock(n + 1, startup) is None
    # combination: integer counters add, the double sums are re-added IN ORDER (not partial sum + partial sum)
    import readsb_amd
    rng = np.random.default_rng(3)
    parts, all_sq, all_terms = [], [], []
    for r in range(3):
        m = np.zeros(1000, dtype=readsb_amd.MSG_DTYPE)
        m["sig_sumsq"] = rng.integers(1, 1 << 40, size=1000, dtype=np.uint64)
        terms = rng.random(50) * 1e-3
        cnt = {k: 0 for k in shard._INT_FIELDS}
        cnt.update(demod_accepted=[r, 1, 0], demod_preamblePhase=[1] * 5, demod_bestPhase=[2] * 5, demod_preambles=10 + r, nflips=99,
                   signal_power_sum=-1.0, noise_power_sum=-1.0, peak_signal_power=0.1 * (r + 1))
        parts.append((m, cnt, terms))
        all_sq.append(m["sig_sumsq"])
        all_terms.append(terms)
    msgs, total = shard.combine_ranges(parts, 131072 * 3 + 5, nflips=7)
    want_sig = 0.0
    for v in np.concatenate(all_sq):
        want_sig += float(v) / 65535.0 / 65535.0
    want_noise = 0.0
    for t in np.concatenate(all_terms):
        want_noise += float(t)
    assert total["signal_power_sum"] == want_sig and total["noise_power_sum"] == want_noise
    assert total["demod_accepted"] == [3, 3, 0] and total["demod_preambles"] == 33 and total["nflips"] == 7
    assert total["peak_signal_power"] == pytest.approx(0.3) and len(msgs) == 3000
    _, total = shard.combine_ranges(parts, 131072 * 3, nflips=7)    # EOF buffer: NaN into the noise sum, one more (lost) buffer
    assert np.isnan(total["noise_power_sum"]) and total["samples_lost"] == 131072 and total["nbuffers"] == 1


def _seq(start, sumsq):
    s = float(start)
    for v in sumsq:
        s += float(v) / 65535.0 / 65535.0
    return s


@pytest.mark.parametrize("case", ["traffic", "ties", "crossings", "tiny-and-huge"])
def test_blockwise_sequential_sum_is_exact(built, case):
    """readsb_amd/csrc/seqsum.cpp: blocks prepared range by range against a PREDICTED binade, applied in O(blocks) by the combining
    rank — bit for bit the one-by-one double sum of the reference (demod_2400.c:445-447), on ordinary traffic, on streams made of
    exact ties (round half to even: every tie's carry is a parity), across binade crossings, and with wrong predictions."""
    import readsb_amd
    from readsb_amd import binding
    rng = np.random.default_rng({"traffic": 1, "ties": 2, "crossings": 3, "tiny-and-huge": 4}[case])
    n, start = 60000, 0.0
    unit = 65535 * 65535                                            # sig_sumsq = k * 65535^2  ->  a signal power of exactly k
    if case == "traffic":
        sumsq = rng.integers(1 << 20, 1 << 42, size=n, dtype=np.uint64)
    elif case == "ties":
        start = float(2 ** 60 + 2 ** 12)                            # grid 256: 128 (mod 256) is half a step
        k = rng.choice([128, 384, 100, 129, 127, 256, 640, 1, 255], size=n)
        sumsq = (k.astype(np.uint64) * np.uint64(unit))
    elif case == "crossings":
        start = 1.0
        sumsq = (rng.integers(0, 4, size=n).astype(np.uint64) * np.uint64(unit)) + rng.integers(0, 1 << 30, size=n, dtype=np.uint64)
    else:
        sumsq = np.where(rng.random(n) < 0.01, rng.integers(1 << 50, 1 << 62, size=n, dtype=np.uint64), rng.integers(0, 50, size=n, dtype=np.uint64)).astype(np.uint64)
    msgs = np.zeros(n, dtype=readsb_amd.MSG_DTYPE)
    msgs["sig_sumsq"] = sumsq
    msgs["msgtype"][::997] = 77                                     # Mode A/C replies in between: no signal power, skipped
    want = _seq(start, sumsq[msgs["msgtype"] != 77])
    assert binding.seqsum_signal_power(start, msgs) == want
    cuts = [0, 7000, 7001, 30000, 52000, n]                         # "ranks": ragged ranges, one of a single message
    for block in (64, 1024):
        # every range predicts from the plain sum of what lies before it (any order, any rounding) ...
        s, fallbacks, nblocks = start, 0, 0
        for a, b in zip(cuts, cuts[1:]):
            approx = start + float(np.sum(sumsq[:a].astype(np.float64) * (msgs["msgtype"][:a] != 77)) / 65535.0 / 65535.0)
            blk = binding.seqsum_blocks(approx, msgs[a:b], block)
            s, fb = binding.seqsum_apply(s, msgs[a:b], blk, block)
            fallbacks += fb
            nblocks += len(blk)
        assert s == want, (case, block, s, want)
        if case in ("traffic", "ties"):
            assert fallbacks < nblocks // 3                          # ... and nearly every block is one integer addition
        # ... and a prediction that is plainly wrong costs speed, never the result
        s = start
        for a, b in zip(cuts, cuts[1:]):
            blk = binding.seqsum_blocks(12345.678, msgs[a:b], block)
            s, _ = binding.seqsum_apply(s, msgs[a:b], blk, block)
        assert s == want


class _ScriptedRank:
    """A rank whose walks are scripted: its warm-up gives the wrong state until it imports; its first clocks are off by a buffer."""

    def __init__(self, rank, world):
        self.rank, self.world, self.import_state, self.walks = rank, world, None, 0
        self.nbuf = 1200

    def _clocks(self, exact):
        c = 1000 + ((np.arange(self.nbuf) + self.rank * self.nbuf) * 131072 * 5) // 12000 + 54
        return c if exact else c - 54

    def estimate(self):
        return self._clocks(False)

    def walk(self, sched):
        self.walks += 1
        true_start = b"" if self.rank == 0 else b"end%d" % (self.rank - 1)
        start = self.import_state if self.import_state is not None else (true_start if self.rank != 1 else b"cold-start-wrong")
        return self._clocks(True), start, b"end%d" % self.rank


def _protocol_worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, helpers.ROOT)
    from readsb_amd import shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    me = _ScriptedRank(rank, world)
    stats = {}
    n = 131072 * 1200 * world + 77
    sched = shard.run_walk_protocol([me], lambda p: shard._all_gather_bytes(p[0], torch.device("cpu")), n, 1000, 0, stats=stats)
    ok = stats["rounds"] == 2 and stats["seam_failures"] == 1 and me.walks == 2 and len(sched) == 3
    ok = ok and (me.import_state == b"end0") == (rank == 1)
    q.put((rank, bool(ok), stats))
    dist.barrier()
    dist.destroy_process_group()


def test_protocol_rounds_over_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 35500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_protocol_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] for r in results), results
