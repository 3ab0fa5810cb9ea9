#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
MGPU_DEVICE_WALK=1 timeout 600 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_device_walk.py 2>&1 | tail -4 > gpurun_out/s60_suite_devwalk.log
cat gpurun_out/s60_suite_devwalk.log
