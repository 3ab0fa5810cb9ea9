#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_golden.py tests/test_gpu_deferred.py tests/test_gpu_device_walk.py tests/test_gpu_modeac.py tests/test_gpu_pipeline_chain.py tests/test_gpu_shard.py -x -q 2>&1 | tail -5
for m in 0 1 0 1; do MGPU_SIG_LATE=$m timeout 200 python bench.py --no-extra-configs --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('sig_late=$m', d['value'], d['ms_per_step'], d.get('stage_ms'))"; done
