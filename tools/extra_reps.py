#!/usr/bin/env python3
"""One extra configuration of bench.py with more repetitions of the timed region: per repetition the rate, the GPU brackets and the host stages.
   usage: MGPU_DBG_BENCH_REPS=5 python tools/extra_reps.py <index> [chunk_buffers] [ahead]"""
import json
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import bench  # noqa: E402
import helpers  # noqa: E402

helpers.ensure_built()
name = list(bench.EXTRA_CONFIGS)[int(sys.argv[1])]
fmt, nfix, kw = bench.EXTRA_CONFIGS[name]
cb = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
ahead = int(sys.argv[3]) if len(sys.argv) > 3 else 2
out = bench.run_extra_config(name, fmt, nfix, kw, 4096 * bench.BUF, 0, chunk_buffers=cb, ahead=ahead)
print(name, out["msamples_s"], out["msamples_s_both_repetitions"])
for r in out["host_stage_ms_both_repetitions"]:
    print("   ", r)
