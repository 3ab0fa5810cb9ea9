#!/bin/bash
cd /root/repo
python - <<'PY'
import bench, json, os, time
import numpy as np
import helpers, readsb_amd
name="dense bursts, 8000 frames/s, overlapping DF17, --aggressive"
fmt, nfix, kw = bench.EXTRA_CONFIGS[name]
nsamples=2048*131072
iq = helpers.synth(nsamples=nsamples, fmt=fmt, seed=424242, threads=64, **kw)
d = readsb_amd.Demodulator(fmt=fmt, nfix_crc=nfix, max_samples=nsamples, device=0, startup_time_ms=helpers.STARTUP_MS)
d.upload_iq(iq)
d.keep_other_threads_away(confine_to_own_l3=False)
d.feed_resident(nsamples); m0,_=d.collect(reuse=True)
bufs=[np.empty(len(m0)*5//4+1024, dtype=readsb_amd.MSG_DTYPE) for _ in range(2)]
for rep in range(4):
    d.reset(); d.set_deferred(True)
    def submit(k):
        d.set_message_buffer(bufs[k%2]); d.feed_resident(nsamples)
    submit(0); d.collect_feed(bufs[0], want_counters=True)
    steps=4
    t0=time.perf_counter(); submit(1)
    ts=[]
    for k in range(2, steps+1):
        submit(k); d.collect_feed(bufs[(k-1)%2]); ts.append(time.perf_counter())
    d.collect_feed(bufs[steps%2], want_counters=True)
    el=time.perf_counter()-t0
    tm=d.timing()
    print(rep, round(el/steps*1e3,3), "ms/segment", {k: round(tm[k],3) for k in ("resolve_ms","build_ms","sigpower_ms","d2h_ms","total_ms")}, [round((b-a)*1e3,2) for a,b in zip([t0]+ts, ts)])
    d.set_deferred(False)
PY
