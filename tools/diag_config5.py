#!/usr/bin/env python3
"""Where the time of a long dense-burst capture goes: per-feed wall times and stage times of the unsharded pass and of the two shard passes.
   usage: tools/diag_config5.py [nsamples]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import torch
import helpers
import readsb_amd
from readsb_amd.shard import run_shard_pass_resident
BUF = 131072
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384 * BUF
n -= n % BUF
helpers.ensure_built()
iq = helpers.synth(nsamples=n, seed=5150, rate=8000.0, dense=1, threads=64)
d_iq = torch.from_numpy(iq).to("cuda:0")
piece = 4096 * BUF
d = readsb_amd.Demodulator(nfix_crc=2, max_samples=piece, device=0, startup_time_ms=helpers.STARTUP_MS)
d.keep_other_threads_away(confine_to_own_l3=False)
bufs = [np.empty(int(piece // 64 + 65536), dtype=readsb_amd.MSG_DTYPE) for _ in range(2)]
offs = list(range(0, n, piece))
for rep in range(2):
    d.reset(); d.set_deferred(True)
    marks = [time.perf_counter()]; tim = []
    for k, off in enumerate(offs):
        d.set_message_buffer(bufs[k % 2])
        d.feed_resident(min(piece, n - off), d_iq.data_ptr() + off * 2)
        if k >= 1:
            m, _ = d.collect_feed(bufs[(k - 1) % 2]); tim.append(d.timing())
        marks.append(time.perf_counter())
    m, _ = d.collect_feed(bufs[(len(offs) - 1) % 2], want_counters=True); tim.append(d.timing())
    marks.append(time.perf_counter())
    d.finish(); d.collect_feed(bufs[0], want_counters=True); d.set_deferred(False)
    print("unsharded rep", rep, "total ms", round((marks[-1] - marks[0]) * 1e3, 2), "per feed ms", [round((b - a) * 1e3, 2) for a, b in zip(marks, marks[1:])])
    for t in tim[:3] + tim[-1:]:
        print("   ", {k: (round(v, 3) if isinstance(v, float) else v) for k, v in t.items()})
for rep in range(2):
    t0 = time.perf_counter()
    bm = run_shard_pass_resident(d, None, 0, n, 1, None, (0, d_iq.data_ptr()), None)
    t1 = time.perf_counter()
    pk = run_shard_pass_resident(d, None, 0, n, 2, bm, (0, d_iq.data_ptr()), None)
    t2 = time.perf_counter()
    d.walk_own_packets()
    t3 = time.perf_counter()
    d.finish(); res = d.collect()
    t4 = time.perf_counter()
    print("sharded rep", rep, "pass1", round((t1 - t0) * 1e3, 2), "pass2", round((t2 - t1) * 1e3, 2), "walk+build", round((t3 - t2) * 1e3, 2), "collect", round((t4 - t3) * 1e3, 2),
          "packet MB", round(pk.size / 1e6, 1), "msgs", len(res[0]))
    print("   ", {k: (round(v, 3) if isinstance(v, float) else v) for k, v in d.timing().items()})
