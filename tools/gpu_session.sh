#!/bin/bash
cd $GRAFT_REPO_ROOT
tag=${1:-sess}
mkdir -p gpurun_out/$tag
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_large.py tests/test_gpu_shard.py tests/test_gpu_prescreen_chains.py tests/test_gpu_formats.py -x -q -m gpu 2>&1 | tail -3
for rep in 1 2; do
  timeout 300 python bench.py --no-cpu-baseline --no-extra-configs 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print(d['value'], d['stage_ms'], d['kernels']['k_slice']['avg_launch_ms'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])"
done 2>&1 | tee gpurun_out/$tag/ab.txt
STAGES=0 bash tools/slice_stage_pmc.sh ${tag}_sl | cut -c1-400
