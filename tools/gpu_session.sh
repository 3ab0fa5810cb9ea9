#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/s24
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15
timeout 600 python bench.py --no-extra-configs > gpurun_out/s24/bench.log 2>&1
tail -1 gpurun_out/s24/bench.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())
print(d['value'], d['ms_per_step'], d['stage_ms'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['kernels'])"
