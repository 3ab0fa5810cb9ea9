#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-configs"
show() { python - "$1" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; c=d['config']
        print(sys.argv[1], round(d['value']), d['ms_per_step'], 'sweep', round(r.get('avg_launch_ms',0),4), round(r['frac'],4), 'stage', c.get('stage_ms'))
PY
}
export MGPU_LIBRARY=libmodes_gpu_exp.so
MGPU_CONV_AHEAD=0 timeout 300 $B > gpurun_out/e_a0.log 2>&1; show gpurun_out/e_a0.log
MGPU_CONV_AHEAD=1 timeout 300 $B > gpurun_out/e_a1.log 2>&1; show gpurun_out/e_a1.log
MGPU_CONV_AHEAD=0 timeout 300 $B > gpurun_out/e_a0b.log 2>&1; show gpurun_out/e_a0b.log
MGPU_CONV_AHEAD=1 timeout 300 $B > gpurun_out/e_a1b.log 2>&1; show gpurun_out/e_a1b.log
MGPU_CONV_AHEAD=1 MGPU_BUILD_THREADS=8 timeout 300 $B > gpurun_out/e_a1t8.log 2>&1; show gpurun_out/e_a1t8.log
MGPU_CONV_AHEAD=1 timeout 300 $B --chunk-buffers 2048 > gpurun_out/e_a1c2048.log 2>&1; show gpurun_out/e_a1c2048.log
MGPU_CONV_AHEAD=0 timeout 300 $B --chunk-buffers 2048 > gpurun_out/e_a0c2048.log 2>&1; show gpurun_out/e_a0c2048.log
unset MGPU_LIBRARY
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/e_full.log 2>&1; tail -c 3000 gpurun_out/e_full.log
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_formats.py tests/test_gpu_pipeline_chain.py -m gpu -x -q 2>&1 | tail -5
