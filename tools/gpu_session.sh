#!/bin/bash
# scratch: what the last GPU session of the round ran (gpurun -- 'bash tools/gpu_session.sh')
cd /root/repo
timeout 600 python bench.py > gpurun_out/bench_final.log 2>&1; tail -1 gpurun_out/bench_final.log | cut -c1-120
