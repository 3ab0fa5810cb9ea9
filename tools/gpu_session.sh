#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-configs"
show() { python - "$1" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print(sys.argv[1], round(d['value']), d['ms_per_step'], 'sweep', round(r.get('avg_launch_ms',0),4), round(r['frac'],4), 'stage', d.get('stage_ms'))
PY
}
timeout 900 python -m pytest tests/test_gpu_prescreen_chains.py tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_kernel_generations.py -m gpu -x -q 2>&1 | tail -5
timeout 300 $B > gpurun_out/p1.log 2>&1; show gpurun_out/p1.log
timeout 300 $B > gpurun_out/p2.log 2>&1; show gpurun_out/p2.log
