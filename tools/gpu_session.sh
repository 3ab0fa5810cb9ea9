#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_golden.py tests/test_gpu_deferred.py tests/test_gpu_pipeline_chain.py -x -q 2>&1 | tail -3
for v in "X=1" "MGPU_SIG_LATE=0" "X=1" "MGPU_SIG_LATE=0"; do env $v timeout 200 python bench.py --no-extra-configs --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'], d.get('stage_ms'))"; done
