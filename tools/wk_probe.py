#!/usr/bin/env python3
"""scratch: one capture through the library with the walk on the device (for rocprofv3 --kernel-trace)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import helpers, readsb_amd
os.environ["MGPU_DEVICE_WALK"] = "1"
iq = helpers.synth(seconds=170.0, seed=77, rate=2400.0, threads=16)
d = readsb_amd.Demodulator(startup_time_ms=helpers.STARTUP_MS, max_samples=len(iq) // 2, nfix_crc=1)
got, cnt = d.demodulate_capture(iq)
print(len(got), d.device_walk_stats())
d.close()
