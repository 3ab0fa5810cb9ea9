#!/usr/bin/env python3
"""bench.py — whole-job throughput of the Mode-S demodulator hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--samples S]

A "step" = one pass of the hot path (IQ -> magnitude -> preamble sweep + bit slicer + CRC ->
ordered accept walk -> signal power) over one synthetic UC8 2.4 MSps stream that is already
resident in HBM when the timed region starts (BASELINE.json configs[1]: single UC8 stream,
--fix).  With N > 1 every rank demodulates its own independent stream (configs[3]) and the
decoded message counts/records are gathered over RCCL inside the timed step.

Prints ONE JSON line (rank 0): metric/value/unit as BASELINE.json, plus
  roofline     — k_sweep_uc8 (UC8 converter and preamble sweep in one kernel, round 6): algorithmic bytes (2 B of IQ read + 2 B of
                 magnitude written per sample) / its HIP-event time; with the two kernels (SC16 input, Mode A/C) k_sweep: 2 B per
                 magnitude sample.  `kernels` carries the same figure for k_slice (slicer + CRC + scoring, the longer of the two)
  cpu_baseline — the reference's own C files (oracle/_ref) timed on this host on the same stream,
                 whose message list must be bit-identical to the GPU's (checked in the same run).
"""
import os
# The HIP runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).  With torch's and RCCL's
# streams in the process too, the library's main stream and its second stream (window statistics, device message build) ended up
# on ONE queue and ran back to back: 429 us per chunk instead of 343 (profiles/r02_queue_sharing.txt).  Must be set before HIP starts.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

BUF = 131072
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)
SWEEP_BYTES_PER_SAMPLE = 2.0   # k_preamble_sweep reads one u16 magnitude per position
FUSED_BYTES_PER_SAMPLE = 4.0   # k_sweep_uc8 (round 6: converter and sweep in one kernel): 2 B of UC8 IQ read + 2 B of magnitude written per sample


# VALU issue rate measured on this chip (tools/micro/valu_issue.hip, profiles/r02_valu_issue.txt): three-operand / packed /
# dot2 instructions — what these kernels are made of — sustain 36 T lane-ops/s chip-wide (v_add/v_xor: ~60)
VALU_PEAK_TLANEOPS = 37.3
N_SIMDS = 1024                 # 256 CUs x 4
PMC_SQ_SUMMARY = os.path.join(ROOT, "profiles", "r06_pmc_sq_summary.txt")
PMC_HBM = os.path.join(ROOT, "profiles", "r06_pmc_hbm.json")


# the launch the committed PMC / SQ summaries were collected on: one chunk of this benchmark's default (2048 buffers = 268 435 456 samples) of UC8 magnitudes
PROFILED_LAUNCH_BYTES = 536870912    # (of magnitudes: 268 435 456 samples per launch)


def _proc_stat():
    out = {}
    for ln in open("/proc/stat"):
        if ln.startswith("cpu") and ln[3].isdigit():
            f = ln.split()
            v = list(map(int, f[1:]))
            out[int(f[0][3:])] = (sum(v), v[3] + v[4], v[0] + v[1], v[2] + v[5] + v[6], v[7] if len(v) > 7 else 0)   # total, idle, user, system + irq, steal
    return out


def host_probe_begin(d):
    """MGPU_DBG_BENCH_HOST=1 (a diagnosis of the processes whose host stages run slow, profiles/r06_host_slow_mode.txt): what the timed
    region's CPUs did — the pipeline's pinned cores, their SMT siblings, the rest of their L3 groups — and where the process's memory is."""
    cpus = d.host_cpus()

    def cpulist(path):
        out = set()
        try:
            for part in open(path).read().strip().split(","):
                lo, _, hi = part.partition("-")
                out.update(range(int(lo), int(hi or lo) + 1))
        except (OSError, ValueError):
            pass
        return out
    sib, l3 = set(), set()
    for c in cpus:
        sib |= cpulist(f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list")
        l3 |= cpulist(f"/sys/devices/system/cpu/cpu{c}/cache/index3/shared_cpu_list")
    freq = {}
    for c in cpus:
        try:
            freq[c] = int(open(f"/sys/devices/system/cpu/cpu{c}/cpufreq/scaling_cur_freq").read())
        except (OSError, ValueError):
            pass
    return {"cpus": list(cpus), "sib": sorted(sib - set(cpus)), "l3": sorted(l3 - sib - set(cpus)), "stat": _proc_stat(), "freq0": freq, "t": time.perf_counter()}


def host_probe_end(p):
    b, a = _proc_stat(), p["stat"]

    def busy(cs):
        rows = []
        for c in cs:
            if c in a and c in b and b[c][0] > a[c][0]:
                tot = b[c][0] - a[c][0]
                rows.append((round(1.0 - (b[c][1] - a[c][1]) / tot, 2), round((b[c][3] - a[c][3]) / tot, 2), round((b[c][4] - a[c][4]) / tot, 2)))
        return rows
    pinned = busy(p["cpus"])
    sib = busy(p["sib"])
    l3 = busy(p["l3"])
    freq = {}
    for c in p["cpus"]:
        try:
            freq[c] = int(open(f"/sys/devices/system/cpu/cpu{c}/cpufreq/scaling_cur_freq").read())
        except (OSError, ValueError):
            pass
    nodes = {}
    try:
        for ln in open("/proc/self/numa_maps"):
            for tok in ln.split():
                if tok[0] == "N" and "=" in tok and tok[1:tok.index("=")].isdigit():
                    ps = 4096
                    if "kernelpagesize_kB=" in ln:
                        ps = 1024 * int(ln.split("kernelpagesize_kB=")[1].split()[0])
                    nodes[tok[1:tok.index("=")]] = nodes.get(tok[1:tok.index("=")], 0) + int(tok.split("=")[1]) * ps
    except OSError:
        pass
    other = [c for c in b if c not in p["cpus"]]
    top_other = sorted(((round(1.0 - (b[c][1] - a[c][1]) / max(1, b[c][0] - a[c][0]), 2), c) for c in other if c in a), reverse=True)[:12]
    return {"seconds": round(time.perf_counter() - p["t"], 3), "pinned_cpus": p["cpus"],
            "pinned_busy_sys_steal": pinned, "sibling_busy_max": max([r[0] for r in sib], default=None), "siblings_busy": [r[0] for r in sib],
            "rest_of_l3_busy_max": max([r[0] for r in l3], default=None),
            "busiest_other_cpus": top_other, "pinned_freq_khz_before": p["freq0"], "pinned_freq_khz_after": freq,
            "process_memory_bytes_by_numa_node": nodes, "loadavg": open("/proc/loadavg").read().split()[:3]}


def kernel_source_sha():
    """Hash of the device code the library was built from (kernels.hip + kernels/*.inc + kernels.h).  The committed PMC
    summaries under profiles/ carry the hash they were collected with (tools/profile_round.sh): numbers of other code are
    not reported as this run's (`traffic` / `valu_issue` stay null)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    base = os.path.join(ROOT, "readsb_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(base, "kernels", "*.inc")) + [os.path.join(base, "kernels.hip"), os.path.join(base, "kernels.h")]):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def valu_issue(kernel, avg_launch_ms, samples_per_launch, path=PMC_SQ_SUMMARY):
    """How busy the vector ALUs are (these kernels are integer-VALU work): VALU wave-instructions of one launch
    from the committed SQ counter pass of this same command (SQ_INSTS_VALU) over the launch time measured live.
    Informative; None if the summary is missing or does not parse."""
    try:
        insts, others, inside, sha_ok = None, 0, False, False
        for ln in open(path):
            if ln.startswith("# kernel_source_sha:"):
                sha_ok = ln.split(":")[1].strip() == kernel_source_sha()
                continue
            if not ln.startswith(" "):
                inside = ln.strip().split("<")[0] in (kernel, kernel + "_t")      # k_sweep is the template k_sweep_t<fused?>
            elif inside and ln.split()[0] == "SQ_INSTS_VALU":
                insts = int(ln.split()[1])
            elif inside and ln.split()[0] in ("SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM"):
                others += int(ln.split()[1])
        if not insts or avg_launch_ms <= 0 or not sha_ok:
            return None
        tl = insts * 64 / (avg_launch_ms * 1e-3) / 1e12
        # Round 5: what bounds these kernels is instruction ISSUE, whatever the instruction — a SIMD issues one long-encoded vector
        # (VOP3 / VOP3P / DPP), scalar or LDS instruction per ~1.8 ns from two resident waves on, a short-encoded VOP2 per ~1.0 ns,
        # and with six waves a vector and a scalar instruction beside each other (profiles/r05_valu_issue.txt): the launch time per
        # wave-instruction of ANY kind and SIMD is the number to hold against those (branches and waits are not counted)
        all_insts = insts + others
        return {"wave_insts_per_launch": insts, "lane_ops_per_sample": round(insts * 64 / samples_per_launch, 1),
                "achieved_Tlaneops_s": round(tl, 2), "peak_Tlaneops_s": VALU_PEAK_TLANEOPS,
                "frac": round(tl / VALU_PEAK_TLANEOPS, 3),
                "all_wave_insts_per_launch": all_insts, "ns_per_inst_per_simd": round(avg_launch_ms * 1e6 * N_SIMDS / all_insts, 2),
                "ns_per_inst_reference": "1.8 long-encoded vector / scalar / LDS alone, 1.0 VOP2 alone, 0.95 a vector + scalar pair at six waves",
                "source": "SQ_INSTS_VALU / SALU / LDS / VMEM / SMEM, " + os.path.relpath(path, ROOT)}
    except Exception:
        return None


def emit(out):
    """The ONE JSON line, as the last line of stdout: RCCL writes its banner ("Librccl path : ...") through C stdio, which is
    flushed at exit — behind Python's line unless it is flushed first."""
    try:
        C.CDLL(None).fflush(None)
    except Exception:
        pass
    print(json.dumps(out), flush=True)


def cpu_reference(iq, nsamples, fmt=0, nfix=1, fixdf=1, thr=58):
    """oracle/_ref (the reference's own convert.c + demodulate2400 + ...) on the host, 1 core."""
    import helpers
    so = os.path.join(ROOT, "oracle", "_ref", "libreadsb_ref.so")
    if os.path.exists(so):
        lib = C.CDLL(so)
        lib.ref_demod_run.argtypes = [C.c_int] * 4 + [C.c_void_p, C.c_uint64, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64),
                                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        out, nout = C.c_void_p(), C.c_uint64()
        st = np.zeros(1, dtype=helpers.ORACLE_STATS)
        devnull = os.open(os.devnull, os.O_WRONLY)
        saved = os.dup(2)
        os.dup2(devnull, 2)   # init_converter prints to stderr
        try:
            rc = lib.ref_demod_run(fmt, nfix, fixdf, thr, iq.ctypes.data, nsamples, C.byref(out), C.byref(nout), st.ctypes.data,
                                   None, None, None)
        finally:
            os.dup2(saved, 2)
            os.close(devnull)
        assert rc == 0
        msgs = np.zeros(nout.value, dtype=helpers.ORACLE_MSG)
        C.memmove(msgs.ctypes.data, out.value, nout.value * helpers.ORACLE_MSG.itemsize)
        return "reference", msgs, st[0]
    msgs, st = helpers.oracle_run(iq[: nsamples * helpers.FMT_BYTES[fmt]], fmt, nfix, fixdf, thr)
    return "port", msgs, st


# The other BASELINE configurations and input statistics, measured in the same run (a point is not a curve: the sweep's and the
# slicer's cost depends on how many positions pass the preamble tests): name -> (format, nfix_crc, synth arguments)
EXTRA_CONFIGS = {
    "configs[2]: SC16Q11 --aggressive (2-bit repair)": (2, 2, dict(rate=2000.0)),
    "dense bursts, 8000 frames/s, overlapping DF17, --aggressive": (0, 2, dict(rate=8000.0, dense=1)),
    "UC8 --fix, Gaussian noise (sigma 3 LSB)": (0, 1, dict(rate=2000.0, dense=4)),
}
# (rounds 2-3 also ran the UC8 stream with the ordered walk on the device, MGPU_DEVICE_WALK=1: 78 Gsamples/s on the driver's box against
# 223 on the builder's, the same code — since round 4 that walk lives in the experiments build only and the product has one walk,
# the host's)
EXTRA_ENV = {}


try:
    ORIG_AFFINITY = os.sched_getaffinity(0)
except (AttributeError, OSError):
    ORIG_AFFINITY = None


def restore_affinity():
    """A context pins its pipeline to cores it picks among those the CALLING thread may run on, and keep_other_threads_away() then takes
    those cores (and their SMT siblings) away from every other thread of the process, the calling one included.  A context created
    later by the same thread would pick among what is left — cores of other L3 groups, further from the device (measured: the
    extra configurations' host walk 2 x slower than in a process of their own) — so the masks go back before the next context."""
    if ORIG_AFFINITY is None:
        return
    for tid in os.listdir("/proc/self/task"):
        try:
            os.sched_setaffinity(int(tid), ORIG_AFFINITY)
        except OSError:
            pass


def run_extra_config(name, fmt, nfix, kw, nsamples, device, steps=32, bracket_us=None, chunk_buffers=0, ahead=1):
    """One more configuration on a fresh context: `steps` back-to-back segments of a resident stream with deferred feeds (timed),
    then two more deferred segments of a fresh stream, fed the same way, whose messages and counters must equal the reference's
    own code on the same two-segment stream."""
    import helpers
    import readsb_amd
    iq = helpers.synth(nsamples=nsamples, fmt=fmt, seed=424242, threads=min(64, os.cpu_count() or 8), **kw)
    restore_affinity()                          # (the previous context of this process is closed: its cores are free again)
    env = EXTRA_ENV.get(name, {})
    saved = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        d = readsb_amd.Demodulator(fmt=fmt, nfix_crc=nfix, max_samples=nsamples, device=device, startup_time_ms=helpers.STARTUP_MS, chunk_buffers=chunk_buffers)
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    d.upload_iq(iq)
    d.keep_other_threads_away(confine_to_own_l3=False)
    d.feed_resident(nsamples)
    m0, _ = d.collect(reuse=True)
    A = max(1, min(3, ahead))                                # segments enqueued ahead of the one being collected
    NB = A + 1
    bufs = [np.empty(len(m0) * 5 // 4 + 1024, dtype=readsb_amd.MSG_DTYPE) for _ in range(NB)]
    d.reset()
    d.set_deferred(True)

    def submit(k):
        d.set_message_buffer(bufs[k % NB])
        d.feed_resident(nsamples)

    seg = [0]

    def run_region(nseg):
        """nseg segments of the stream, A of them in flight beyond the one being collected; ends with an empty pipeline."""
        first = seg[0]
        for k in range(first, first + nseg + A):
            if k < first + nseg:
                submit(k)
            if k - A >= first:
                d.collect_feed(bufs[(k - A) % NB], want_counters=(k - A == first + nseg - 1))
        seg[0] += nseg

    # warm-up: the timed region's own protocol, untimed (every job and slot of the pipeline at full size — ten slots, twelve jobs, a
    # segment is two to four chunks — and the kernels' pace estimates settled on THIS configuration's step times: with two drained
    # segments of warm-up the first repetition came out 10-20 % under the later ones, five repetitions in a row, and with 16 segments
    # — 40 ms — a walk team still in its first 100 ms made one first repetition 3.4 instead of 1.0 ms per segment, gpurun r06aq): three
    # regions' worth, ~0.25 s, as the headline's own warm-up
    run_region(3 * steps)
    # the timed region, twice: both rates are reported and the entry's figure is their MEAN (round 4 took the better one; the pool's
    # boxes are shared nodes — load average 14-21 while these ran — and a host stage of one repetition now and then runs slower:
    # that is part of what a deployment sees); the stage times are the slower repetition's
    runs = []
    for _ in range(int(os.environ.get("MGPU_DBG_BENCH_REPS", "2"))):
        d.timing()
        t0 = time.perf_counter()
        run_region(steps)
        runs.append((time.perf_counter() - t0, d.timing()))
    elapsed = sum(r[0] for r in runs) / len(runs)
    tm = max(runs, key=lambda r: r[0])[1]
    if bracket_us is None:
        bracket_us = d.event_bracket_us()       # what a pair of timing events adds to what it brackets (see main())
    ev_scale = tm["n_chunks"] / max(1, tm["n_timed_chunks"])
    for key in ("convert_ms", "sweep_ms", "slice_ms", "prescreen_ms"):
        tm[key] *= ev_scale
    # the check: the same deferred feeds from a fresh stream — two segments, the consumer's arrays, feed k+1 enqueued before feed k
    # is collected, exactly what was timed — against the reference's own code on the two-segment stream
    d.reset()
    got = []
    submit(0)
    submit(1)
    m, _ = d.collect_feed(bufs[0])
    got.append(m.copy())
    m, counters = d.collect_feed(bufs[1 % NB], want_counters=True)
    got.append(m.copy())
    d.finish()
    _, counters = d.collect_feed(bufs[0], want_counters=True)
    d.set_deferred(False)
    msgs = np.concatenate(got)
    iq2 = np.concatenate([iq, iq])
    if helpers.have_ref():                                   # the reference's objects hold one configuration per process: own process
        kind, (ref_msgs, st) = "reference", helpers.ref_run(iq2, fmt, nfix, 1, 58)
    else:
        kind, (ref_msgs, st) = "port", helpers.oracle_run(iq2, fmt, nfix, 1, 58)
    del iq2
    helpers.assert_same_messages(msgs, ref_msgs)
    helpers.assert_same_counters(counters, st)
    nl = max(1, tm["n_chunks"])
    out = {"msamples_s": round(nsamples * steps / elapsed / 1e6, 1), "ms_per_segment": round(elapsed / steps * 1e3, 3),
           "msamples_s_both_repetitions": [round(nsamples * steps / r[0] / 1e6, 1) for r in runs],
           "host_stage_ms_both_repetitions": [{k2: round(r[1][k1] / steps, 3) for k1, k2 in (("d2h_ms", "d2h"), ("resolve_ms", "resolve_host"), ("build_ms", "build_host"), ("build_wait_ms", "build_wait"), ("sigpower_ms", "sigpower"))} for r in runs],
           "samples_per_segment": nsamples, "segments_timed": steps, "chunk_buffers": chunk_buffers or 1024, "segments_ahead": A, "messages_per_segment": int(len(msgs)),
           "candidates_per_1000_samples": round(tm["n_candidates"] / (nsamples * steps) * 1e3, 2),
           "records_per_1000_samples": round(tm["n_records"] / (nsamples * steps) * 1e3, 2),
           "live_records_per_1000_samples": round(tm["n_live_records"] / (nsamples * steps) * 1e3, 2),
           # (per launch, the events' own constant taken off; stage_ms below: as the events report it)
           "us_per_launch": {"convert": round(tm["convert_ms"] / nl * 1e3 - bracket_us, 1), "k_sweep": round(tm["sweep_ms"] / nl * 1e3 - bracket_us, 1),
                             "k_slice": round(tm["slice_ms"] / nl * 1e3 - bracket_us, 1), "post_sweep": round(tm["prescreen_ms"] / nl * 1e3 - bracket_us, 1)},
           "event_bracket_us": round(bracket_us, 2),
           "samples_per_launch": int(nsamples * steps // nl),
           # per segment: the GPU stages (HIP events, summed over the segment's launches) and the host stages behind them (wall
           # clock of the fetcher / walker / builder threads: they overlap each other and the GPU)
           "stage_ms": {k2: round(tm[k1] / steps, 3) for k1, k2 in (("convert_ms", "convert"), ("sweep_ms", "sweep"), ("slice_ms", "slice"),
                                                                   ("prescreen_ms", "prescreen"), ("d2h_ms", "d2h"), ("resolve_ms", "resolve_host"),
                                                                   ("build_ms", "build_host"), ("sigpower_ms", "sigpower"))},
           "cpu_reference_msamples_s": round(2 * nsamples / float(st["t_convert_s"] + st["t_demod_s"]) / 1e6, 1),
           "bit_identical_to_reference": True, "checked": "two deferred segments of a fresh stream, fed as in the timed region", "checker": kind}
    d.close()
    return out


def bench_config5(args, rank, local_rank, world):
    """BASELINE configs[4]: ONE dense-burst capture (overlapping 112-bit DF17 frames, --aggressive) time-chunked by whole buffers
    over the ranks, every rank walking and building its OWN range (readsb_amd/shard.py: ShardWalkRank / run_walk_protocol; round 4
    — rounds 2-3 shipped every range's records to rank 0, which walked the whole capture alone).  Strong scaling: the capture is
    fixed, a step = the whole capture once; every rank's samples (its range + two filter generations of warm-up) are resident in
    its HBM (--samples 8640000000 = the one-hour capture of BASELINE.json, 17.3 GB, fits one MI355X).
      N = 1: the capture through the ordinary pipeline (one stream, deferred feeds) — with one rank sharding IS that;
      N > 1: the protocol over torch.distributed (RCCL, or gloo with --dryrun-gloo);
      --emulate-ranks K on ONE GPU: one context plays K ranks one after the other, every rank's phases timed on their own, the
      all-gathers replaced by lists: what each rank would spend, what the combining rank does on top of its own share, and
      the result checked against the unsharded run and the reference like any other."""
    import torch
    import torch.distributed as dist
    import helpers
    import readsb_amd
    from readsb_amd import shard
    from readsb_amd.shard import shard_ranges, needed_from, warmup_start
    dev = 0 if args.dryrun_gloo else local_rank
    torch.cuda.set_device(dev)
    coll = torch.device("cpu") if args.dryrun_gloo else torch.device("cuda", local_rank)
    if world > 1 and not dist.is_initialized():
        if args.dryrun_gloo:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    helpers.ensure_built()
    n = args.samples - args.samples % BUF
    emu = args.emulate_ranks if world == 1 else 0
    first, last = shard_ranges(n, world)[rank]
    lo = needed_from(first)
    threads = min(64, max(1, (os.cpu_count() or 8) // world))
    t_g0 = time.time()
    mine = helpers.synth(nsamples=last - lo, first=lo, seed=5150, rate=8000.0, dense=1, threads=threads)   # the rank's range, its warm-up and histories
    t_gen = time.time() - t_g0
    d_iq = torch.from_numpy(mine).to(torch.device("cuda", dev))            # resident: its own allocation, any length
    piece = min(16384 * BUF, max(BUF, ((last - first) // max(1, args.emulate_ranks if world == 1 else 1) // BUF + 512) * BUF))   # samples per feed call: a rank's range in one
    d = readsb_amd.Demodulator(nfix_crc=2, max_samples=piece, device=dev, startup_time_ms=helpers.STARTUP_MS, chunk_buffers=args.chunk_buffers)
    resident = (lo, d_iq.data_ptr())
    d.keep_other_threads_away(confine_to_own_l3=world > 1)

    def hist_of(sample):
        return None if sample == 0 else mine[(sample - 326 - lo) * 2:(sample - lo) * 2].copy()

    # ---- the capture through the ordinary pipeline (one stream, deferred feeds of `piece` samples, resident IQ): N = 1's value,
    #      and the time everything else is measured against ----
    def unsharded_pass(keep):
        bufs = [np.empty(int(piece // 64 + 65536), dtype=readsb_amd.MSG_DTYPE) for _ in range(2)]
        offs = list(range(0, n, piece))
        d.reset()
        d.set_deferred(True)
        got, nm = [], 0
        t1 = time.perf_counter()
        for k, off in enumerate(offs):
            d.set_message_buffer(bufs[k % 2])
            d.feed_resident(min(piece, n - off), d_iq.data_ptr() + off * 2)
            if k >= 1:
                m, _ = d.collect_feed(bufs[(k - 1) % 2])
                nm += len(m)
                if keep:
                    got.append(m.copy())
        m, _ = d.collect_feed(bufs[(len(offs) - 1) % 2], want_counters=True)
        nm += len(m)
        if keep:
            got.append(m.copy())
        dt = time.perf_counter() - t1
        d.finish()                                   # end of file: ifileRun's last, empty buffer (sdr_ifile.c:223-237)
        _, cnt = d.collect_feed(bufs[0], want_counters=True)
        d.set_deferred(False)
        d.set_message_buffer(None)
        return dt, nm, (np.concatenate(got) if keep else None), cnt

    out = {"metric": "IQ Msamples/s demodulated, one dense-burst UC8 capture time-chunked over the GPUs (--aggressive), whole job",
           "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "higher_is_better": True, "scaling": "strong",
           "vs_baseline": None, "dtype": "u16", "data": "synthetic", "synth_gen_s": round(t_gen, 2)}
    workload = (f"configs[4]: one {n / 2.4e6:.1f} s dense-burst UC8 capture ({n} samples, 8000 overlapping DF17 frames/s, --aggressive) "
                f"time-chunked by whole buffers over {world} GPU(s), each range (+ {shard.WARMUP / 2.4e6:.1f} s of warm-up before it) resident in its GPU's HBM; ")
    msgs = counters = None
    um = uc = None
    if world == 1:
        for _ in range(max(0, args.warmup)):
            unsharded_pass(False)
        torch.cuda.synchronize()
        best, total = None, 0.0
        for _ in range(max(1, args.steps)):              # timed: the consumer takes each feed's messages where they are built
            dt, nm, _, _ = unsharded_pass(False)
            total += dt
            best = dt if best is None or dt < best else best
        _, _, um, uc = unsharded_pass(True)              # checked: the same feeds, their messages kept
        msgs, counters = um, uc
        elapsed = total
        out.update(value=round(n * max(1, args.steps) / elapsed / 1e6, 1), ms_per_step=round(elapsed / max(1, args.steps) * 1e3, 3), messages_per_step=int(len(um)),
                   best_step_ms=round(best * 1e3, 3))
        out["config"] = {"workload": workload + "one rank: the ordinary pipeline over the whole capture (deferred feeds)", "samples": n, "shards": 1, "parallelism": "time-chunked x1"}
    else:
        stats, phases = {}, {}
        wf = warmup_start(first)
        hists = {wf: hist_of(wf), first: hist_of(first)}

        def gather_ranges(ranges):                   # the pre-pass's buffers, gathered into one device buffer (HBM -> HBM)
            t = torch.cat([d_iq[(a - lo) * 2:(b - lo) * 2] for a, b in ranges])
            return t, t.data_ptr()

        def one_pass(ph=None, st=None):
            if args.config5_form == "stream":
                return shard.demodulate_sharded_stream(d, None, coll, resident=resident, nsamples=n, histories=hists, gather=gather_ranges, phases=ph, stats=st,
                                                       concat=False)
            return shard.demodulate_sharded_walk(d, None, coll, resident=resident, nsamples=n, histories=hists, phases=ph, stats=st)

        for _ in range(max(0, args.warmup)):
            one_pass()
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = None
        for _ in range(args.steps):
            res = one_pass(phases, stats)
        dist.barrier()
        torch.cuda.synchronize()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=coll)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        if rank == 0:
            msgs, counters = res
            if isinstance(msgs, list):               # (the stream form hands the ranges' arrays over as they arrived; the checks below want one)
                msgs = np.concatenate(msgs)
            out.update(value=round(n * args.steps / elapsed / 1e6, 1), ms_per_step=round(elapsed / args.steps * 1e3, 3), messages_per_step=int(len(msgs)),
                       rank0_phase_ms_per_step={k: round(v / args.steps, 3) for k, v in phases.items()}, protocol=stats)
            out["config"] = {"workload": workload + f"every rank walks and builds its own range ({args.config5_form} form); per round an all-gather of every buffer's "
                                                    "end clock and the filter state at every range's ends; at the end the ranges' messages gathered on rank 0",
                             "samples": n, "shards": world, "parallelism": f"time-chunked x{world}"}

    # ---- K emulated ranks on this one GPU: what each would spend, what the combining rank adds ----
    if emu > 1:
        stats = {}
        t_ser = {}

        def lap(name, t0):
            t_ser[name] = t_ser.get(name, 0.0) + (time.perf_counter() - t0) * 1e3

        # every rank's standing message array (an aggregator keeps one; page faults of a fresh one are not the rank's work)
        outs = [np.zeros(int((b - a) // 512 + 65536), dtype=readsb_amd.MSG_DTYPE) for a, b in shard_ranges(n, emu)]
        startup, fc = int(d.cfg.startup_time_ms), int(d.cfg.filter_clock)
        stream = args.config5_form == "stream"
        if stream:
            def gather(ranges):                      # the pre-pass's buffers, gathered into one device buffer (HBM -> HBM)
                t = torch.cat([d_iq[(a - lo) * 2:(b - lo) * 2] for a, b in ranges])
                return t, t.data_ptr()
            src = shard._Source(d, resident=resident, histories=None, gather=gather)
            src.history = lambda sample: hist_of(sample)
        rounds = 0
        for trial in range(2):                       # the first time round warms the context's buffers up (its numbers are dropped)
            t_ser.clear()
            if stream:
                ranks = [shard.ShardStreamRank(d, r, emu, n, src, out=outs[r]) for r in range(emu)]
                for r in ranks:
                    r.terms_view = True              # (a real rank has its own context: its signal terms stay where the builder logged them)
                pre = [r.prepass(startup, fc) for r in ranks]
                t0 = time.perf_counter()
                sched = shard.schedule_from_window_estimates(pre, n, startup, fc)
                lap("schedule_from_estimates", t0)
                step = lambda r, sc: r.stream_pass(sc)
            else:
                ranks = [shard.ShardWalkRank(d, r, emu, n, keep_packets=True, out=outs[r]) for r in range(emu)]
                for r in ranks:
                    r.gpu_phase(None, resident=resident, histories={r.ws: hist_of(r.ws)})
                t0 = time.perf_counter()
                sched = shard.schedule_from_clocks([r.estimate() for r in ranks], n, startup, fc)
                lap("schedule_from_estimates", t0)
                step = lambda r, sc: r.walk(sc)
            rounds = 0
            while True:
                rounds += 1
                got = []
                for r in ranks:
                    got.append(shard._unpack(shard._pack(*step(r, sched))))
                    if stream:
                        # the rank's sum blocks, from its signal terms while they are where the builder logged them (the emulation's one
                        # context is about to play the next rank); a real rank prepares them behind the counters' all-gather — the time is
                        # the rank's either way
                        t0 = time.perf_counter()
                        r.blocks = shard.prepare_sum_blocks(r.msgs, [q.counters for q in ranks[:r.rank]], r.sig_terms)
                        r.ms["sum_blocks"] = (time.perf_counter() - t0) * 1e3
                        r.sig_terms = None
                t0 = time.perf_counter()
                done, nxt, imports = shard.protocol_round(sched, got, n, startup, fc)
                lap("round_conclusions", t0)
                if trial == 1:
                    stats["seam_failures"] = stats.get("seam_failures", 0) + len(imports)
                if done:
                    break
                sched = nxt
                for r in ranks:
                    if r.rank in imports:
                        r.import_state = imports[r.rank]
                assert rounds < emu + 72
        for r in ranks:
            if stream:
                continue
            t0 = time.perf_counter()
            r.blocks = shard.prepare_sum_blocks(r.msgs, [q.counters for q in ranks[:r.rank]], getattr(r, "sig_terms", None))
            r.ms["sum_blocks"] = (time.perf_counter() - t0) * 1e3
        cstats = {}
        t0 = time.perf_counter()
        # (the concatenation of the message arrays stands in for the gather's receive side: RCCL lands them in rank 0's memory)
        parts = [(r.msgs, r.counters, r.noise, r.blocks) for r in ranks]
        t1 = time.perf_counter()
        elist, ecnt = shard.combine_ranges(parts, n, len(sched) + (1 if fc == 1 else 0), cstats, concat=False)
        t_comb = (time.perf_counter() - t1) * 1e3
        t_ser["combine_counters_and_sums"] = t_comb
        emsgs = np.concatenate(elist)               # (the check below wants one array; an aggregator's gather lands the ranges in one buffer)
        assert um is not None and len(emsgs) == len(um) and emsgs.tobytes() == um.tobytes(), "emulated ranks and the unsharded run differ"
        helpers.assert_same_counters(ecnt, uc)
        per_rank = [{k: round(v, 3) for k, v in r.ms.items()} for r in ranks]
        # a rank's critical path: GPU phase, clock estimate | round(s): walk + build + collect | its sum blocks; the combining rank's
        # extra: conclusions per round, the combination (the gather itself is communication: bytes below)
        crit = [r.ms.get("prepass", 0) + r.ms.get("stream_pass", 0) + r.ms.get("gpu_phase", 0) + r.ms.get("clock_estimate", 0) + r.ms.get("walk_call", 0) + r.ms.get("collect", 0) + r.ms.get("sum_blocks", 0) for r in ranks]
        serial = sum(t_ser.values())
        unsh_ms = out["ms_per_step"]
        out["emulated_ranks"] = {
            "ranks": emu, "form": args.config5_form, "per_rank_ms": per_rank, "rank_critical_path_ms": [round(c, 3) for c in crit],
            "protocol": {"rounds": rounds, "walks": [r.walks for r in ranks], "seam_failures": stats.get("seam_failures", 0), "imported": [r.import_state is not None for r in ranks],
                         "expiries": int(len(sched)), "sum_blocks": cstats.get("sum_blocks"), "sum_blocks_readded": cstats.get("sum_blocks_readded")},
            "rank0_serial_ms": {k: round(v, 3) for k, v in t_ser.items()}, "rank0_serial_total_ms": round(serial, 3),
            "rank0_serial_share_of_unsharded": round(serial / unsh_ms, 4),
            "projected_ms_without_communication": round(max(crit) + serial, 3),
            "projected_speedup_without_communication": round(unsh_ms / (max(crit) + serial), 2),
            "gather_bytes_to_rank0": int(sum(r.msgs.nbytes + r.blocks.nbytes + r.noise.nbytes for r in ranks[1:])),
            "allgather_bytes_per_round": int(sum(len(shard._pack(*g)) for g in got)),
            "identical_to_unsharded": True, "note": "one context plays the ranks one after the other on ONE GPU: every rank's phases are what it would spend "
                                                    "with the GPU to itself; the all-gathers are lists, the final gather is not timed (bytes given)"}

    if rank == 0:
        if not args.no_cpu_baseline:
            iq = helpers.synth(nsamples=n, seed=5150, rate=8000.0, dense=1, threads=min(64, os.cpu_count() or 8)) if (world > 1 or lo != 0 or last != n) else mine
            kind, ref_msgs, st = cpu_reference(iq, n, 0, 2, 1, 58)
            helpers.assert_same_messages(msgs, ref_msgs)
            helpers.assert_same_counters(counters, st)            # every counter, the order-dependent double sums included
            cpu_s = float(st["t_convert_s"] + st["t_demod_s"])
            out["cpu_baseline"] = {"value": round(n / cpu_s / 1e6, 1), "unit": "Msamples/s", "cores": 1, "kind": kind,
                                   "sample": f"the whole capture ({n} samples) on one host core: convert {st['t_convert_s']:.1f} s + demodulate2400 {st['t_demod_s']:.1f} s",
                                   "messages": int(len(ref_msgs)), "bit_identical_to_gpu": True, "counters_compared": "all"}
        emit(out)
    d.close()
    if dist.is_initialized():
        dist.destroy_process_group()


def relaunch_or_check_world(args):
    """`--gpus N` means N ranks, one per GPU.  Under torchrun (WORLD_SIZE set) the flag and the environment must agree.  Without
    torchrun and N > 1 this process becomes the launcher: it runs itself under `python -m torch.distributed.run --nnodes=1
    --nproc-per-node N --master-addr 127.0.0.1` with the same arguments and exits with the launcher's status — so
    `python bench.py --gpus 8` and the driver's explicit torchrun command are the same job.  More ranks than GPUs is an error,
    never a silent one-GPU run (round 5: the flag was parsed and ignored)."""
    import subprocess
    if args.gpus < 1:
        raise SystemExit(f"bench.py: --gpus {args.gpus}: at least one")
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is not None and int(env_world) != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher's WORLD_SIZE is {env_world}: pass --gpus {env_world} "
                         f"(or start bench.py without torchrun and let --gpus launch the ranks)")
    if args.gpus > 1 and not args.dryrun_gloo:
        import torch
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but this node shows {have} GPU(s): one rank per GPU "
                             f"(--dryrun-gloo runs the N-rank code path on one device)")
    if env_world is None and args.gpus > 1:
        import socket
        with socket.socket() as s:                              # a free rendezvous port on the loopback interface
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.stdout.flush()
        raise SystemExit(subprocess.call(cmd))


def launch_check(args):
    """--launch-check: the ranks find each other and agree on the world size; rank 0 prints it.  With --dryrun-gloo no GPU is touched."""
    import torch
    import torch.distributed as dist
    world, local_rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))
    seen = 1
    if world > 1:
        if args.dryrun_gloo:
            dist.init_process_group("gloo")
            t = torch.ones(1, dtype=torch.int64)
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            t = torch.ones(1, dtype=torch.int64, device=torch.device("cuda", local_rank))
        dist.all_reduce(t)
        seen = int(t.item())
        dist.destroy_process_group()
    if int(os.environ.get("RANK", "0")) == 0:
        emit({"launch_check": True, "n_gpus": world, "ranks_seen": seen, "gpus_flag": args.gpus,
              "launched_by": "torchrun" if os.environ.get("TORCHELASTIC_RUN_ID") else "direct"})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--samples", type=int, default=4096 * BUF, help="samples of the HBM-resident IQ block = of one feed call (multiple of 131072)")
    ap.add_argument("--loops", type=int, default=50, help="feeds per step: a step plays the resident block this many times (the stream goes on: an ifile "
                    "in a loop).  50 x 537 M samples = 26.8 G samples per step: the default 20 steps are ~1.1 s of timed region, not 28 ms")
    ap.add_argument("--ahead", type=int, default=2, help="feeds enqueued ahead of the one being collected (1..3: the library keeps at most four uncollected).  "
                    "2: with chunks of 2048 buffers a feed of 4096 is two chunks, and fetcher, walker and builder want a chunk each to work on while "
                    "the GPU runs the next (1: 1.48 ms per feed, 2: 1.12-1.17, 3: 1.12-1.13; profiles/r06_chunk_2048.txt)")
    ap.add_argument("--device-build", action="store_true", help="N = 1: messages built by k_build_messages and copied into the consumer's page-locked "
                    "array (mgpu_set_device_messages(2)) instead of by the host's builder threads.  Measured slower (profiles/r06_device_build.txt): "
                    "the host builder is not what bounds the step, and the 4.7 MB per chunk of message copies queue ahead of the fetcher's record copies")
    ap.add_argument("--msgs-per-sec", type=float, default=2000.0)
    ap.add_argument("--config", type=int, default=1, help="1 (default): one stream per GPU (BASELINE configs[1] / configs[3]); 5: one dense-burst "
                                                          "capture time-chunked over the GPUs (configs[4], strong scaling)")
    ap.add_argument("--emulate-ranks", type=int, default=0, help="--config 5 on ONE GPU: one context plays this many ranks of the sharded walk one after "
                                                                  "the other; per-rank phase times and the combining rank's serial share in the JSON line")
    ap.add_argument("--chunk-buffers", type=int, default=None, help="mgpu_config.chunk_buffers: buffers per pipeline chunk (0 = the library's 1024).  Default: 2048 "
                    "for --config 1 — half the launches and kernel tails per sample (k_slice 0.46 against 0.52 ms per 537 M samples), the host's walk "
                    "takes a chunk in rounds of 1024 buffers whatever its length (profiles/r06_chunk_2048.txt); 0 for --config 5")
    ap.add_argument("--event-bracket-us", type=float, default=None, help="what a pair of timing events adds to the kernel it brackets, as measured by an "
                    "earlier run (`event_bracket_us` of its line): skips the calibration (k_spin launches) — for rocprofv3 runs, whose kernel statistics "
                    "then hold the pipeline's kernels only")
    ap.add_argument("--config5-form", choices=["stream", "packets"], default="stream",
                    help="--emulate-ranks: `stream` = every rank's pass through the ordinary pipeline, the schedule from a pre-pass over the expiry "
                         "windows (walk and build overlap the kernels); `packets` = GPU pass first, then the walk of its packets")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-configs", action="store_true", help="skip the `configs` dict (other BASELINE configurations / input statistics)")
    ap.add_argument("--extra-samples", type=int, default=4096 * BUF, help="samples per segment of the extra configurations (default: the headline step)")
    ap.add_argument("--main-cpu", type=int, default=-1, help="experiment: pin the calling thread to this CPU after the context exists")
    ap.add_argument("--exercise-gather", action="store_true", help="run the N>1 aggregator exchange even with one rank (needs torchrun env)")
    ap.add_argument("--dryrun-gloo", action="store_true",
                    help="dry run of the N>1 code path on ONE GPU: gloo backend, collectives on CPU tensors, every rank on device 0 "
                         "(torchrun --nproc-per-node 2 bench.py --gpus 2 --dryrun-gloo ...); the numbers mean nothing")
    ap.add_argument("--launch-check", action="store_true",
                    help="the N-rank launch alone (no GPU needed with --dryrun-gloo): every rank joins the process group, rank 0 prints a line "
                         "with the world size it found — tests/test_bench_launch.py")
    ap.add_argument("--extra-only", type=int, default=-1, help="(internal) run the i-th extra configuration alone and print {name: result}")
    ap.add_argument("--extra-device", type=int, default=0, help="(internal) the device of --extra-only")
    args = ap.parse_args()
    if args.chunk_buffers is None:
        args.chunk_buffers = 2048 if args.config == 1 else 0
    if args.extra_only >= 0:
        import helpers
        helpers.ensure_built()
        name = list(EXTRA_CONFIGS)[args.extra_only]
        fmt, nfix, kw = EXTRA_CONFIGS[name]
        try:
            res = run_extra_config(name, fmt, nfix, kw, args.extra_samples - args.extra_samples % BUF, args.extra_device, bracket_us=args.event_bracket_us,
                                   chunk_buffers=args.chunk_buffers, ahead=max(1, min(3, args.ahead)))
        except AssertionError as e:
            sys.stderr.write(str(e)[-1500:] + "\n")
            raise SystemExit(3)
        emit({name: res})
        return

    relaunch_or_check_world(args)
    if args.launch_check:
        return launch_check(args)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    n = args.samples - args.samples % BUF
    assert n > 0

    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: libmodes_gpu has no CPU path")
    if args.config == 5:
        return bench_config5(args, rank, local_rank, world)
    if args.dryrun_gloo:
        local_rank_dev, coll_dev = 0, torch.device("cpu")
    else:
        local_rank_dev, coll_dev = local_rank, torch.device("cuda", local_rank)
    torch.cuda.set_device(local_rank_dev)
    use_dist = world > 1 or args.exercise_gather
    if use_dist:
        if args.dryrun_gloo:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import helpers
    import readsb_amd
    from readsb_amd.gather import MessageGatherer
    helpers.ensure_built()

    # ---- synthetic input: one independent stream per rank, generated on the host, staged in HBM ----
    t0 = time.time()
    threads = max(1, (os.cpu_count() or 8) // max(1, world))
    iq = helpers.synth(nsamples=n, seed=88172645463325252 + rank, rate=args.msgs_per_sec, threads=min(threads, 64))
    t_gen = time.time() - t0
    d = readsb_amd.Demodulator(max_samples=n, device=local_rank_dev, startup_time_ms=helpers.STARTUP_MS, chunk_buffers=args.chunk_buffers)
    d.upload_iq(iq)
    # the application's threads (this one, the HIP / RCCL runtime's) stay off the pipeline's cores; with several ranks on the
    # node they stay inside the rank's own CCD, on the SMT siblings the pipeline leaves free
    d.keep_other_threads_away(confine_to_own_l3=world > 1)
    if args.main_cpu >= 0:
        os.sched_setaffinity(0, {args.main_cpu})

    gath = None                                 # N > 1: the aggregator role, asynchronous (readsb_amd/gather.py)
    A = max(1, min(3, args.ahead))              # feeds in flight beyond the one being collected

    # ---- sizing pass (synchronous): the consumer's standing message arrays, 1.25 x the busiest rank's message count ----
    d.reset()
    d.feed_resident(n)
    m0, _ = d.collect(reuse=True)
    cap = len(m0) * 5 // 4 + 1024
    if use_dist:
        t = torch.tensor([cap], dtype=torch.int64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        cap = int(t.item())
        gath = MessageGatherer(readsb_amd.MSG_DTYPE, coll_dev, cap, depth=A + 2, host_alloc=d.host_alloc)   # (a slot per feed in flight + the one being gathered)
        bufs = None
    else:
        # the consumer's two standing arrays, page-locked near the device: k_build_messages stores the records into them (mode 2)
        nbufs = 1 + A
        bufs = [np.empty(cap, dtype=readsb_amd.MSG_DTYPE) for _ in range(nbufs)] if not args.device_build else [d.host_alloc(cap * 64).view(readsb_amd.MSG_DTYPE)[:cap] for _ in range(nbufs)]
        if os.environ.get("MGPU_DBG_PINNED_BUFS"):
            bufs = [(d.host_alloc(cap * 64) if os.environ["MGPU_DBG_PINNED_BUFS"] == "near" else torch.empty(cap * 64, dtype=torch.uint8, pin_memory=True).numpy()).view(readsb_amd.MSG_DTYPE) for _ in range(2)]

    # ---- the job: ONE continuous stream, step k = its k-th 223.7 s segment (the same resident IQ block again: an ifile played
    #      in a loop), fed with deferred calls — feed(k+1) is enqueued before the messages of feed(k) are taken, so the pipeline
    #      does not run empty between steps; a step's messages are built straight into the consumer's array (no copy) ----
    d.reset()
    d.set_deferred(True)
    # N > 1: the messages are built on the GPU and go from HBM into the RCCL gather — no host build, no staging upload
    dev_msgs = gath is not None and not args.dryrun_gloo and not os.environ.get("MGPU_DBG_HOST_MESSAGES")
    if dev_msgs:
        d.set_device_messages(True)
    # N = 1, --device-build: the GPU builds the records too and copies them into the consumer's page-locked arrays — the host's
    # builder stage keeps the two order-dependent sums (round 6; measured slower than the builder threads: not the default)
    dev_to_host = gath is None and args.device_build
    if dev_to_host:
        d.set_device_messages(2)
    arrays = {}

    dbg_t = {"staging": 0.0, "feed": 0.0, "collect": 0.0, "gsubmit": 0.0}

    def submit(k):
        t_a = time.perf_counter()
        buf = None
        if dev_msgs:
            d.set_device_message_buffer(*gath.device_buffer(ahead=k - gath.seq))   # the feed's records are built where the gather sends them from
        if not dev_msgs:
            buf = gath.staging(ahead=k - gath.seq) if gath is not None else bufs[k % len(bufs)]
            d.set_message_buffer(buf)
        t_b = time.perf_counter()
        d.feed_resident(n)                      # everything from HBM-resident IQ to ordered messages
        arrays[k] = buf
        dbg_t["staging"] += t_b - t_a
        dbg_t["feed"] += time.perf_counter() - t_b

    def take(k, want_counters=False):
        t_a = time.perf_counter()
        if dev_msgs:
            dptr, nmsgs, counters = d.collect_feed_device(want_counters=want_counters)
            arrays.pop(k)
            t_b = time.perf_counter()
            gath.submit_inplace(nmsgs)          # count + records from this rank's HBM to rank 0's over RCCL (one collective), overlapped with the next feeds
        else:
            msgs, counters = d.collect_feed(arrays.pop(k), want_counters=want_counters)
            nmsgs = len(msgs)
            t_b = time.perf_counter()
            if gath is not None:
                gath.submit(nmsgs)              # (dry run / MGPU_DBG_HOST_MESSAGES: through the pinned staging ring)
        dbg_t["collect"] += t_b - t_a
        dbg_t["gsubmit"] += time.perf_counter() - t_b
        return nmsgs, counters

    seq = 0
    L = max(1, args.loops)
    n_warm, n_timed = args.warmup * L, args.steps * L          # feeds

    def run_feeds(first, count):
        """`count` feeds from number `first` on, A of them enqueued ahead of the one being collected; ends with an empty pipeline."""
        last = None
        for k in range(count + A):
            if k < count:
                submit(first + k)
            if k >= A:
                last = take(first + k - A, want_counters=(k - A == count - 1))
        return last

    if n_warm:
        run_feeds(seq, n_warm)                                  # drained: the timed region starts with an empty pipeline
        seq += n_warm
    d.timing()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    dbg_host = host_probe_begin(d) if os.environ.get("MGPU_DBG_BENCH_HOST") else None
    t0 = time.perf_counter()
    nmsgs_last, counters = run_feeds(seq, n_timed)              # ... and ends with an empty one
    if use_dist:
        gath.wait()                             # the last steps' exchanges are part of the job
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dbg_host is not None:
        sys.stderr.write("dbg host " + json.dumps(host_probe_end(dbg_host)) + "\n")
    if os.environ.get("MGPU_DBG_BENCH"):
        sys.stderr.write(f"dbg bench rank {rank}: main thread, warm-up + timed steps: {dbg_t}\n")
    tm = d.timing()                             # sums over the timed region's launches (everything since the last drain)
    # what two timing events around one kernel report beyond the kernel (include/modes_gpu.h): measured here, or taken from an earlier run
    bracket_us = args.event_bracket_us if args.event_bracket_us is not None else d.event_bracket_us()
    # the stage events ride on every 15th chunk only (an event costs ~5 us of idle stream): scale the sampled sums to all chunks
    ev_scale = tm["n_chunks"] / max(1, tm["n_timed_chunks"])
    for key in ("convert_ms", "sweep_ms", "slice_ms", "prescreen_ms"):
        tm[key] *= ev_scale
    K = float(n_timed)                         # the stage figures below are per FEED (one pass over the resident block), as in rounds 1-5
    launches = [max(1, tm["n_chunks"]) / K]
    sweep_ms, slice_ms, conv_ms = [tm["sweep_ms"] / K], [tm["slice_ms"] / K], [tm["convert_ms"] / K]
    resolve_ms, total_ms = [tm["resolve_ms"] / K], [tm["total_ms"] / K]
    for key in ("prescreen_ms", "d2h_ms", "build_ms", "sigpower_ms", "build_wait_ms"):
        tm[key] = tm[key] / K
    d.set_deferred(False)                       # (also leaves the device-messages mode)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        nm = torch.tensor([nmsgs_last], dtype=torch.int64, device=coll_dev)
        dist.all_reduce(nm)
        total_msgs = int(nm.item())
        # every rank's host stages (wall clock of the work itself, per step): do N pipelines' polling threads get in each other's way?
        hs = torch.tensor([resolve_ms[0], tm["build_ms"], tm["d2h_ms"], len(d.host_cpus()), tm["build_wait_ms"]], dtype=torch.float64, device=coll_dev)
        hparts = [torch.zeros_like(hs) for _ in range(world)]
        dist.all_gather(hparts, hs)
        per_rank_host = [{"resolve_host": round(float(x[0]), 3), "build_host": round(float(x[1]), 3), "d2h": round(float(x[2]), 3), "pinned_cpus": int(x[3]), "build_wait": round(float(x[4]), 3)} for x in hparts]
    else:
        total_msgs = nmsgs_last
        per_rank_host = None

    # ---- bit-identity + CPU baseline, same run.  Every rank checks its own stream: the first two segments, fed exactly as in
    #      the timed region (deferred, into the consumer's arrays), against the reference's own code on the 2-segment stream, on
    #      one host core per stream — with N streams that is N pinned reference processes side by side ----
    cpu_result = None
    if not args.no_cpu_baseline:
        d.reset()
        d.set_deferred(True)
        vb = bufs if bufs is not None and dev_to_host else [np.empty(cap, dtype=readsb_amd.MSG_DTYPE) for _ in range(2)]
        if dev_to_host:
            d.set_device_messages(2)                            # checked as timed: the GPU's records, out of the page-locked arrays
        for k in range(2):
            d.set_message_buffer(vb[k])
            d.feed_resident(n)
        g0, _ = d.collect_feed(vb[0])
        g1, vcnt = d.collect_feed(vb[1], want_counters=True)
        d.finish()
        _, vcnt = d.collect_feed(vb[0], want_counters=True)
        gpu_msgs = np.concatenate([g0, g1])
        d.set_deferred(False)
        iq2 = np.concatenate([iq[: n * 2], iq[: n * 2]])
        saved_affinity = None
        try:                                                    # a core of its own, off the pipeline's cores and their siblings
            saved_affinity = os.sched_getaffinity(0)
            free = sorted(saved_affinity)
            os.sched_setaffinity(0, {free[(7 + 2 * local_rank) % len(free)]})
        except (OSError, AttributeError, ZeroDivisionError):
            saved_affinity = None
        if use_dist:
            dist.barrier()                                      # the N reference processes run at the same time
        kind, ref_msgs, st = cpu_reference(iq2, 2 * n)
        del iq2
        if saved_affinity:
            os.sched_setaffinity(0, saved_affinity)            # (threads created from here on inherit it)
        cpu_s = float(st["t_convert_s"] + st["t_demod_s"])
        helpers.assert_same_messages(gpu_msgs, ref_msgs)        # bit-identical decoded message set, same run
        helpers.assert_same_counters(vcnt, st)
        rate = 2 * n / cpu_s / 1e6
        if world > 1:
            t = torch.tensor([rate, cpu_s], dtype=torch.float64, device=coll_dev)
            parts = [torch.zeros_like(t) for _ in range(world)]
            dist.all_gather(parts, t)
            rates = [float(x[0].item()) for x in parts]
            slowest = max(float(x[1].item()) for x in parts)
        else:
            rates, slowest = [rate], cpu_s
        cpu_result = {"value": round(world * 2 * n / slowest / 1e6, 1), "unit": "Msamples/s", "cores": world, "kind": kind,
                      "sample": f"the first two steps of every rank's stream ({2 * n} samples each), {world} reference process(es) side by side, "
                                f"one pinned host core each: rank 0 convert {st['t_convert_s']:.2f} s + demodulate2400 {st['t_demod_s']:.2f} s "
                                f"({os.cpu_count()} cores present)",
                      "per_stream_msamples_s": [round(r, 1) for r in rates],
                      "messages": int(len(ref_msgs)), "bit_identical_to_gpu": True}

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        ms_per_feed = elapsed / n_timed * 1e3
        value = n * world * n_timed / elapsed / 1e6
        nlaunch = int(np.mean(launches))
        # Per step: sum over the step's launches.  A launch's figure is a pair of timing HIP events around the one kernel; in a busy
        # stream that pair reports the kernel + a constant (4.0 us here: mgpu_event_bracket_us measures it in this process with a
        # kernel of known duration), which is taken off — rocprofv3's kernel durations (profiles/r03_kernel_stats.csv) are the check.
        sweep_raw, slice_raw = float(np.mean(sweep_ms)), float(np.mean(slice_ms))
        sweep = max(sweep_raw - nlaunch * bracket_us * 1e-3, 1e-6)
        slice_ = max(slice_raw - nlaunch * bracket_us * 1e-3, 1e-6)
        # `achieved` / `frac`: from the RAW interval between the two events around the kernel (what rocprofv3's average duration of the
        # same kernel agrees with); the figure with the events' own constant taken off stands beside it (`frac_bracket_corrected`)
        if sweep_raw <= 0 or slice_raw <= 0:     # (cannot happen: the first chunk of every accounting period is a sampled one)
            raise SystemExit("bench.py: no chunk of the timed region carried the stage timing events")
        # The dominant kernel: since round 6 the preamble sweep converts on the way (k_sweep_uc8: the UC8 samples read once, the
        # magnitudes written once for k_slice — the converter's launch and the sweep's own read of the magnitudes are gone), so its
        # algorithmic bytes are the converter's (2 B read + 2 B written per sample); a context that runs the two
        # kernels (experiments build, MGPU_SWEEP_FUSED=0; the SC16 formats; Mode A/C) reports k_sweep with 2 B per sample as before.
        fused = tm.get("sweep_fused_chunks", 0) > 0
        bytes_per_sample = FUSED_BYTES_PER_SAMPLE if fused else SWEEP_BYTES_PER_SAMPLE
        sweep_name = "k_sweep_uc8" if fused else "k_sweep"
        achieved = n * bytes_per_sample / (sweep_raw * 1e-3) / 1e9
        achieved_corrected = n * bytes_per_sample / (sweep * 1e-3) / 1e9
        per_launch = int(n * SWEEP_BYTES_PER_SAMPLE / nlaunch)           # (the launch's size in bytes of magnitudes: what the committed summaries were taken on)
        alg_per_launch = int(n * bytes_per_sample / nlaunch)
        # HBM traffic of one launch from the committed rocprofv3 PMC passes of this same command (FETCH_SIZE x2 on
        # gfx950 + WRITE_SIZE, tools/pmc_summary.py); null if the launch size differs
        traffic, traffic_slice = None, None
        try:
            pm = json.load(open(PMC_HBM))
            if per_launch == PROFILED_LAUNCH_BYTES and pm.get("kernel_source_sha") == kernel_source_sha():
                traffic = round(next(v for k, v in pm.items() if ("mgpu::" + sweep_name + "(") in k or k.rstrip() == "mgpu::" + sweep_name)["hbm_bytes"])
                traffic_slice = round(next(v for k, v in pm.items() if "mgpu::k_slice" in k)["hbm_bytes"])
        except Exception:
            pass
        out = {
            "metric": "IQ Msamples/s demodulated (UC8 2.4 MSps stream, --fix), whole job",
            "value": round(value, 1), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u16", "data": "synthetic",
            "config": {"workload": "configs[1]: single 2.4 MSps UC8 stream per GPU, --fix (nfix_crc=1, fixDF=1, thr=58), "
                                   f"one continuous stream, a step = {L} feeds of the resident IQ block ({n} samples each) = {n * L} samples = {n * L / 2.4e6:.0f} s of it, "
                                   f"{args.msgs_per_sec:.0f} frames/s, HBM-resident IQ, pipeline chunks of {args.chunk_buffers or 1024} buffers, deferred feeds ({A} enqueued ahead of the one being collected), "
                                   + ("messages built on the GPU and stored into the consumer's page-locked arrays" if dev_to_host else "messages built on the GPU, gathered from HBM" if dev_msgs else "messages built by the host's builder threads"),
                       "samples_per_stream": n * L, "samples_per_feed": n, "feeds_per_step": L, "streams": world, "parallelism": f"1 stream per GPU x{world}"},
            "ms_per_feed": round(ms_per_feed, 3),
            "x_realtime_per_gpu": round(value / world / 2.4, 1),
            "msgs_per_s": round(total_msgs * n_timed / elapsed, 1),
            "messages_per_step": total_msgs * L, "messages_per_feed": total_msgs,
            "stage_ms": {"convert": round(float(np.mean(conv_ms)), 3), "sweep": round(sweep_raw, 3), "slice": round(slice_raw, 3),
                         "prescreen": round(tm["prescreen_ms"], 3), "d2h": round(tm["d2h_ms"], 3),
                         "resolve_host": round(float(np.mean(resolve_ms)), 3), "build_host": round(tm.get("build_ms", 0.0), 3), "build_wait": round(tm.get("build_wait_ms", 0.0), 3), "sigpower": round(tm["sigpower_ms"], 3), "per": "feed",
                         "feed_total": round(float(np.mean(total_ms)), 3)},
            "roofline": {"kernel": sweep_name, "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "launches_per_step": nlaunch, "algorithmic_bytes_per_launch": alg_per_launch,
                         "algorithmic_bytes": ("2 B of UC8 IQ read + 2 B of magnitude written per sample (converter and sweep in one kernel)" if fused
                                               else "2 B of magnitude read per position"),
                         "avg_launch_ms": round(sweep_raw / nlaunch, 4), "avg_launch_ms_bracket_corrected": round(sweep / nlaunch, 4),
                         "frac_bracket_corrected": round(achieved_corrected / HBM_PEAK_GBS, 4),
                         "event_bracket_us": round(bracket_us, 2), "launches_timed": int(tm["n_timed_chunks"]),
                         "valu_issue": valu_issue(sweep_name, sweep / nlaunch, n / nlaunch) if per_launch == PROFILED_LAUNCH_BYTES else None},
            # the other half of what used to be one kernel: slicer + CRC + scoring over k_sweep's candidate lists.  It reads the
            # same 2 B per sample again (tile staging), so the same algorithmic bytes; its work is per candidate, not per byte.
            "kernels": {"k_slice": {"avg_launch_ms": round(slice_raw / nlaunch, 4), "avg_launch_ms_bracket_corrected": round(slice_ / nlaunch, 4),
                                    "algorithmic_bytes_per_launch": per_launch,
                                    "achieved": round(n * SWEEP_BYTES_PER_SAMPLE / (slice_raw * 1e-3) / 1e9, 1) if slice_raw > 0 else None,
                                    "frac": round(n * SWEEP_BYTES_PER_SAMPLE / (slice_raw * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if slice_raw > 0 else None,
                                    "unit": "GB/s", "traffic": traffic_slice,
                                    "valu_issue": valu_issue("k_slice", slice_ / nlaunch, n / nlaunch) if per_launch == PROFILED_LAUNCH_BYTES else None}},
            "synth_gen_s": round(t_gen, 2),
        }
        if per_rank_host is not None:
            out["per_rank_host_ms"] = per_rank_host
        # the boundary handing over HOST buffers (mgpu_feed_iq from page-locked memory): never `value`, see DESIGN.md §4
        try:
            d.host_register(iq)
            rates = []
            for _ in range(3):
                d.reset()
                t1 = time.perf_counter()
                d.feed_iq(iq)
                d.finish()
                pm, _ = d.collect(reuse=True)
                rates.append(n / (time.perf_counter() - t1) / 1e6)
            d.host_unregister(iq)
            out["pcie_inclusive_msamples_s"] = round(max(rates), 1)
        except Exception as e:                                   # the measurement is informative only
            out["pcie_inclusive_msamples_s"] = None
            out["pcie_inclusive_error"] = str(e)[:200]
        if cpu_result is not None:
            out["cpu_baseline"] = cpu_result
        if not args.no_extra_configs and not args.no_cpu_baseline and world == 1:
            d.close()
            out["configs"] = {}
            # Every extra configuration in a process of its own (`--extra-only i`): as the third or fourth context of THIS process the
            # dense-burst configuration came out at 123-248 Gsamples/s in repetitions that a fresh process runs at 288-302, eighteen in a
            # row (gpurun r06as / r06at) — whatever the earlier contexts leave behind (thread placement, the allocator's state) is not the
            # configuration's.  The child checks against the reference exactly as before; exit code 3 = a mismatch.
            import subprocess
            restore_affinity()
            for i, name in enumerate(EXTRA_CONFIGS):
                cmd = [sys.executable, os.path.abspath(__file__), "--extra-only", str(i), "--extra-samples", str(args.extra_samples),
                       "--chunk-buffers", str(args.chunk_buffers), "--ahead", str(A), "--extra-device", str(local_rank_dev)]
                if args.event_bracket_us is not None:
                    cmd += ["--event-bracket-us", str(args.event_bracket_us)]
                try:
                    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
                    if r.returncode == 3:                        # a mismatch is a failed run, not a missing number
                        raise SystemExit(f"extra configuration '{name}': GPU result differs from the reference: {r.stderr[-1500:]}")
                    if r.returncode != 0:
                        raise RuntimeError(r.stderr[-300:])
                    out["configs"].update(json.loads(r.stdout.strip().splitlines()[-1]))
                except SystemExit:
                    raise
                except Exception as e:                           # anything else (an allocation, the checker's binary): this entry is missing, the line is not
                    out["configs"][name] = {"error": f"{type(e).__name__}: {e}"[:300]}
        emit(out)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
