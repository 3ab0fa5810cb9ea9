#!/bin/bash
# scratch: what the last GPU session of the round ran (gpurun -- 'bash tools/gpu_session.sh')
cd /root/repo
timeout 500 python bench.py --config 5 --emulate-ranks 8 --samples 8640000000 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/c5_hour.log 2>&1
tail -1 gpurun_out/c5_hour.log | cut -c1-600
