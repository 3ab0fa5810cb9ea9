#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_golden.py tests/test_gpu_deferred.py -x -q 2>&1 | tail -2
for v in "X=1" "MGPU_SIG_LATE=0" "X=1"; do env $v timeout 200 python bench.py --no-extra-configs --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'], d.get('stage_ms'))"; done
python - <<'PY'
import bench, json, os
for name in ["dense bursts, 8000 frames/s, overlapping DF17, --aggressive"]:
    fmt, nfix, kw = bench.EXTRA_CONFIGS[name]
    for env in ({}, {"MGPU_SIG_LATE": "0"}):
        os.environ.pop("MGPU_SIG_LATE", None); os.environ.update(env)
        r = bench.run_extra_config(name, fmt, nfix, kw, 2048*131072, 0)
        print(env, r["msamples_s"], r["ms_per_segment"], r["us_per_launch"])
PY
