#!/bin/bash
cd /root/repo
for ms in 1 2 1 2; do
MGPU_MAIN_STREAMS=$ms timeout 300 python bench.py --no-extra-configs 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())
s=d['stage_ms']
print('main_streams=$ms', d['value'], d['ms_per_step'], s, 'sweep', d['roofline']['avg_launch_ms'], 'slice', d['kernels']['k_slice']['avg_launch_ms'], 'identical', d.get('cpu_baseline',{}).get('bit_identical_to_gpu'))"
done
MGPU_MAIN_STREAMS=2 timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
