#!/usr/bin/env python3
"""A long continuous stream through the pipeline with different chunk lengths and numbers of feeds in flight: every feed's messages
(hashed) and the final counters must be the same whatever the chunking — kernels' partitioning, the walk's rounds, the slots' reuse and
the deferred protocol all change between the variants, the results may not.  The first variant's first two feeds are also what
bench.py checks against the reference.
   usage: python tools/stress_stream.py [feeds=400] [buffers_per_feed=4096] [dense=0]"""
import hashlib
import os
import sys
import time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np  # noqa: E402
import helpers  # noqa: E402
import readsb_amd  # noqa: E402

helpers.ensure_built()
feeds = int(sys.argv[1]) if len(sys.argv) > 1 else 400
nbuf = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
dense = int(sys.argv[3]) if len(sys.argv) > 3 else 0
n = nbuf * 131072
kw = dict(rate=8000.0, dense=1) if dense else dict(rate=2000.0)
iq = helpers.synth(nsamples=n, seed=20260930, threads=min(64, os.cpu_count() or 8), **kw)
ref = None
for chunk_buffers, ahead in ((2048, 2), (1024, 1), (4096, 3), (512, 3), (1536, 2)):
    d = readsb_amd.Demodulator(max_samples=n, startup_time_ms=helpers.STARTUP_MS, chunk_buffers=chunk_buffers, nfix_crc=2 if dense else 1)
    d.upload_iq(iq)
    d.keep_other_threads_away(confine_to_own_l3=False)
    cap = None
    d.feed_resident(n)
    m0, _ = d.collect(reuse=True)
    cap = len(m0) * 5 // 4 + 1024
    bufs = [np.empty(cap, dtype=readsb_amd.MSG_DTYPE) for _ in range(ahead + 1)]
    d.reset()
    d.set_deferred(True)
    hashes, total = [], 0
    t0 = time.perf_counter()
    for k in range(feeds + ahead):
        if k < feeds:
            d.set_message_buffer(bufs[k % (ahead + 1)])
            d.feed_resident(n)
        if k >= ahead:
            j = k - ahead
            msgs, cnt = d.collect_feed(bufs[j % (ahead + 1)], want_counters=(j == feeds - 1))
            hashes.append(hashlib.sha256(msgs.tobytes()).hexdigest()[:16])
            total += len(msgs)
    dt = time.perf_counter() - t0
    d.set_deferred(False)
    d.close()
    cnt = {k: (v.tolist() if hasattr(v, "tolist") else v) for k, v in cnt.items()}
    print(f"chunks of {chunk_buffers} buffers, {ahead} ahead: {feeds} feeds, {total} messages, {feeds * n / dt / 1e9:.1f} Gsamples/s", flush=True)
    if ref is None:
        ref = (hashes, total, repr(cnt))
    else:
        bad = [i for i, (a, b) in enumerate(zip(ref[0], hashes)) if a != b]
        assert not bad and total == ref[1], f"feeds {bad[:10]} differ from the first variant's ({len(bad)} of {feeds})"
        assert repr(cnt) == ref[2], "counters differ from the first variant's"
print(f"stress_stream ok: 5 variants x {feeds} feeds of {nbuf} buffers identical")
