#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/s62
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 900 python bench.py > gpurun_out/s62/bench.log 2> gpurun_out/s62/bench.err
tail -1 gpurun_out/s62/bench.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())
print(d['value'], d['ms_per_step'], d['stage_ms'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['kernels']['k_slice']['avg_launch_ms'])
for k,v in d.get('configs',{}).items(): print(k, v.get('msamples_s'), v.get('ms_per_segment'))"
