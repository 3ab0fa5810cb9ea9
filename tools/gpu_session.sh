#!/bin/bash
cd $GRAFT_REPO_ROOT
tag=${1:-sess}
mkdir -p gpurun_out/$tag
for rep in 1 2; do
for lib in libmodes_gpu.so libmodes_gpu_wg3.so; do
  MGPU_LIBRARY=$lib timeout 300 python bench.py --no-cpu-baseline --no-extra-configs 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$lib', d['value'], d['stage_ms'], d['kernels']['k_slice']['avg_launch_ms'])"
done
done 2>&1 | tee gpurun_out/$tag/ab.txt
