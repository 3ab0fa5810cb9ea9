#!/bin/bash
cd /root/repo
timeout 900 python bench.py > gpurun_out/bench_default.log 2>&1; tail -1 gpurun_out/bench_default.log | cut -c1-200
timeout 900 python bench.py --steps 20 --warmup 3 > gpurun_out/bench_s20.log 2>&1; tail -1 gpurun_out/bench_s20.log | cut -c1-200
