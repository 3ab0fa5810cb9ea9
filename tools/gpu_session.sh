#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05h
uptime
for rep in 1 2 3 4 5 6 7 8; do
  timeout 300 python bench.py --no-cpu-baseline --no-extra-configs 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print(d['value'], d['stage_ms'])"
done 2>&1 | tee gpurun_out/r05h/runs.txt
timeout 600 python -m pytest tests/test_gpu_deferred.py tests/test_gpu_parity.py tests/test_gpu_dropin.py -x -q -m gpu 2>&1 | tail -2
