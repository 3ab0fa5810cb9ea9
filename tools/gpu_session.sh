#!/bin/bash
cd $GRAFT_REPO_ROOT
tag=${1:-sess}
mkdir -p gpurun_out/$tag
for rep in 1 2; do
for lib in libmodes_gpu.so libmodes_gpu_prews.so libmodes_gpu_r4end.so; do
  for i in 2 1; do
    MGPU_LIBRARY=$lib timeout 300 python tools/profile_extra.py $i 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=list(d)[0]; v=d[k]
print('$lib', k[:20], v.get('msamples_s_both_repetitions'), v.get('us_per_launch'), v.get('host_stage_ms_both_repetitions')[1])"
  done
done
done 2>&1 | tee gpurun_out/$tag/ab.txt
