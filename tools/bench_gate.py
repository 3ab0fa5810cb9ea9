"""Throughput of the first-stage tracking gate (kernels/gate.inc) on message lists resident in HBM: python tools/bench_gate.py.
Two traffic shapes — 200 aircraft (long runs per address) and 30 000 aircraft (short runs) — demodulated on the GPU from seeded
synthetic captures, replicated in time to ~1 M messages.  One JSON line per shape: messages/s of mgpu_track_gate_device (wall clock
around the C-ABI call: prep, 4 x (count, scan, scatter), runs, walk, verdict + stream sync), verdict shares.  Under
`rocprofv3 --kernel-trace --stats` for the kernels' own durations."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers  # noqa: E402
import readsb_amd  # noqa: E402

BUF = 131072


def main():
    hip = C.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipFree.argtypes = [C.c_void_p]
    for label, kw, seconds in (("200 aircraft", dict(seed=4242, rate=2000.0), 30.0), ("30000 aircraft", dict(seed=777, rate=2500.0, naircraft=30000), 30.0)):
        iq = helpers.synth(seconds=seconds, threads=16, **kw)
        nsamp = len(iq) // 2
        nbuf = (nsamp + BUF - 1) // BUF
        d = readsb_amd.Demodulator(startup_time_ms=helpers.STARTUP_MS, max_samples=nsamp)
        msgs, _ = d.demodulate_capture(iq)
        reps = max(1, (1 << 20) // len(msgs))
        parts = []
        for k in range(reps):                                   # the same traffic again, nbuf buffers later
            m = msgs.copy()
            m["timestamp"] += k * nbuf * BUF * 5
            m["sysTimestamp"] += (k * nbuf * BUF * 5) // 12000
            parts.append(m)
        allm = np.ascontiguousarray(np.concatenate(parts))
        n = len(allm)
        d_msgs, d_fields, d_v = C.c_void_p(), C.c_void_p(), C.c_void_p()
        assert hip.hipMalloc(C.byref(d_msgs), allm.nbytes) == 0 and hip.hipMalloc(C.byref(d_fields), n * readsb_amd.FIELDS_DTYPE.itemsize) == 0 and hip.hipMalloc(C.byref(d_v), n) == 0
        assert hip.hipMemcpy(d_msgs, allm.ctypes.data, allm.nbytes, 1) == 0
        d.decode_fields_device(d_msgs.value, n, d_fields.value)
        times = []
        for _ in range(6):
            d.track_gate_reset()
            t0 = time.perf_counter()
            d.track_gate_device(d_msgs.value, d_fields.value, n, d_v.value)
            times.append(time.perf_counter() - t0)
        v = np.empty(n, dtype=np.uint8)
        assert hip.hipMemcpy(v.ctypes.data, d_v, n, 2) == 0
        t = min(times[1:])
        print(json.dumps({"traffic": label, "messages": n, "distinct_addresses": int(len(np.unique(allm["addr"]))), "ms": round(t * 1e3, 3), "messages_per_s": round(n / t),
                          "x_realtime_at_2000_msgs_per_s": round(n / t / 2000.0), "forward": int(((v & 3) == 1).sum()), "drop": int(((v & 3) == 0).sum()),
                          "defer": int(((v & 3) == 2).sum()), "timing": "wall clock around mgpu_track_gate_device, launches + sync included, best of 5"}))
        for p in (d_msgs, d_fields, d_v):
            hip.hipFree(p)
        d.close()


if __name__ == "__main__":
    main()
