#!/bin/bash
# scratch: one GPU session
cd $GRAFT_REPO_ROOT
tag=${1:-sess}
mkdir -p gpurun_out/$tag
for v in 0 1 0 1; do
  MGPU_LIBRARY=libmodes_gpu_exp.so MGPU_CONVERT_BESIDE=$v timeout 300 python bench.py --no-cpu-baseline --no-extra-configs 2>gpurun_out/$tag/err$v.txt | tail -1 > gpurun_out/$tag/bench_$v.json
  python - <<PY
import json
try:
    d=json.load(open('gpurun_out/$tag/bench_$v.json'))
    print('beside=$v', d['value'], d['stage_ms'], d['kernels']['k_slice']['avg_launch_ms'], d['roofline']['avg_launch_ms'])
except Exception as e: print('beside=$v failed', e)
PY
done
MGPU_LIBRARY=libmodes_gpu_exp.so MGPU_CONVERT_BESIDE=1 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_formats.py tests/test_gpu_deferred.py -x -q -m gpu 2>&1 | tail -3
