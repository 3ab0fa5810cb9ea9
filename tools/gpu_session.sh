#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05j
uptime
for rep in 1 2 3; do
  timeout 600 python bench.py > gpurun_out/r05j/d$rep.log 2> gpurun_out/r05j/d$rep.err
  tail -1 gpurun_out/r05j/d$rep.log | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('default', d['value'], d['stage_ms'])"
  timeout 300 python bench.py --no-cpu-baseline --no-extra-configs 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('plain  ', d['value'], d['stage_ms'])"
done 2>&1 | tee gpurun_out/r05j/runs.txt
uptime
