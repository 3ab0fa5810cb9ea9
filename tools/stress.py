"""Shake the host pipeline: many steps back to back, many context lifetimes, odd thread counts."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import helpers, readsb_amd
helpers.ensure_built()
iq = helpers.synth(seconds=30.0, seed=7, threads=16)
want, _ = helpers.oracle_run(iq)
n = len(iq) // 2
t0 = time.time()
for wt, bt in [(4, 3), (1, 1), (2, 5), (7, 2)]:
    os.environ["MGPU_WALK_THREADS"], os.environ["MGPU_BUILD_THREADS"] = str(wt), str(bt)
    for life in range(6):
        d = readsb_amd.Demodulator(max_samples=n, startup_time_ms=helpers.STARTUP_MS)
        d.upload_iq(iq)
        for step in range(40):
            d.reset(); d.feed_resident(n); d.finish()
            msgs, _ = d.collect(reuse=True)
            assert len(msgs) == len(want), (wt, bt, life, step, len(msgs), len(want))
        helpers.assert_same_messages(msgs, want)
        d.close()
print("stress ok: 960 feeds, 24 context lifetimes, %.1f s" % (time.time() - t0))
