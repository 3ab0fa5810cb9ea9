#!/bin/bash
# What one gpurun call of this round usually ran.  (Scratch: edited per call.)
cd /root/repo
O=gpurun_out/r04f
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_formats.py tests/test_gpu_modeac.py tests/test_gpu_parity.py tests/test_gpu_deferred.py tests/test_gpu_golden.py tests/test_gpu_dropin.py -m gpu -x -q 2>&1 | tail -4
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_full.json 2> $O/bench_full.err
python - <<'PY'
import json
try:
    o = json.loads(open("gpurun_out/r04f/bench_full.json").read().strip().splitlines()[-1])
    r = o["roofline"]
    print(round(o["value"]), o["ms_per_step"], "sweep", r["avg_launch_ms"], r["frac"], "raw", r["avg_launch_ms_between_events"], "slice", o["kernels"]["k_slice"]["avg_launch_ms"], o["stage_ms"], "pcie", o.get("pcie_inclusive_msamples_s"))
    for k, v in o.get("configs", {}).items():
        print(k, {kk: v.get(kk) for kk in ("msamples_s", "ms_per_segment", "us_per_launch", "stage_ms", "error")})
except Exception as e:
    print("no line:", e)
PY
tail -3 $O/bench_full.err
