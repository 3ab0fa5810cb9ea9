"""Throughput of the other configurations (not the headline metric): SC16 / SC16Q11 input, 2-bit repair."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import helpers, readsb_amd
helpers.ensure_built()
N = 2048 * 131072
for name, fmt, nfix, ac in [("UC8 nfix=1", 0, 1, 0), ("UC8 nfix=1 + Mode A/C", 0, 1, 1), ("UC8 nfix=2", 0, 2, 0), ("SC16 nfix=1", 1, 1, 0), ("SC16Q11 nfix=2 (config 3)", 2, 2, 0)]:
    iq = helpers.synth(nsamples=N, fmt=fmt, seed=11, threads=32, dense=2 if ac else 0, rate=800.0 if ac else 2000.0)
    d = readsb_amd.Demodulator(fmt=fmt, nfix_crc=nfix, max_samples=N, startup_time_ms=helpers.STARTUP_MS, mode_ac=ac)
    d.upload_iq(iq)
    best = 1e9
    for _ in range(5):
        d.reset(); t = time.perf_counter(); d.feed_resident(N); d.finish(); m, _ = d.collect(reuse=True); best = min(best, time.perf_counter() - t)
    tm = d.timing()
    print(f"{name:28s} {N / best / 1e9:7.1f} Gsamples/s  step {best * 1e3:6.2f} ms  convert {tm['convert_ms']:.2f} sweep {tm['sweep_ms']:.2f} post {tm['prescreen_ms']:.2f} ms  msgs {len(m)}")
    d.close()
