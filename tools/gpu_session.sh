#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_device_walk.py -x -q -s 2>&1 | grep -v "^$" | tail -10
MGPU_DEBUG_PRINT=1 MGPU_DEVICE_WALK=1 timeout 200 python bench.py --no-extra-configs --no-cpu-baseline --steps 3 --warmup 1 2>gpurun_out/s45.err | tail -1 | cut -c1-200
grep "device walk" gpurun_out/s45.err | sed -n 30,36p
for i in 1 2; do MGPU_DEVICE_WALK=1 timeout 200 python bench.py --no-extra-configs --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('walk=1', d['value'], d['ms_per_step'], d.get('stage_ms'))"; done
