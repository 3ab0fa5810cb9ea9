#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/s50
timeout 600 python bench.py --no-extra-configs > gpurun_out/s50/bench.log 2>&1
tail -1 gpurun_out/s50/bench.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())
print(d['value'], d['ms_per_step'], d['stage_ms']); print(d['roofline']); print(d['kernels'])"
