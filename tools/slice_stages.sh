#!/bin/bash
# k_slice with one part left out at a time (experiments build; the results are wrong, the launch time is the point):
#   0 = everything, 1 = no scoring pass, 2 = no group slicing, 3 = no record writes, 5 = stage-in only, 6 = no DF stage (hence no slicing, no scoring)
cd "$(dirname "$0")/.."
for st in 0 1 2 3 5 6; do
  MGPU_LIBRARY=libmodes_gpu_exp.so MGPU_DEBUG_STAGE=$st timeout 300 python bench.py --no-cpu-baseline --no-extra-configs --steps 3 --warmup 1 --loops 4 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('stage $st: k_slice us', round(d['kernels']['k_slice']['avg_launch_ms']*1e3,1), 'k_sweep us', round(d['roofline']['avg_launch_ms']*1e3,1), 'step ms', d['ms_per_step'])"
done
