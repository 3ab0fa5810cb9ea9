#!/bin/bash
# scratch: per-session GPU command
cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_device_walk.py -x -q -s 2>&1 | tail -40 > gpurun_out/s40_devwalk.log
cat gpurun_out/s40_devwalk.log
