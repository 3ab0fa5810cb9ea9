#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_golden.py tests/test_gpu_deferred.py tests/test_gpu_modeac.py tests/test_gpu_shard.py tests/test_gpu_parity.py -x -q 2>&1 | tail -4
for m in 1 1; do timeout 200 python bench.py --no-extra-configs --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('stage_ms'))"; done
