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
export MGPU_LIBRARY=libmodes_gpu_exp.so
for i in 1 2 3; do
MGPU_PUBLISH_ASIDE=0 timeout 300 $B > gpurun_out/a0_$i.log 2>&1; show gpurun_out/a0_$i.log
MGPU_PUBLISH_ASIDE=1 timeout 300 $B > gpurun_out/a1_$i.log 2>&1; show gpurun_out/a1_$i.log
done
unset MGPU_LIBRARY
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_deferred.py tests/test_gpu_shard.py -m gpu -x -q 2>&1 | tail -3
