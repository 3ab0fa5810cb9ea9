#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/s44
timeout 900 python -m pytest tests/test_gpu_convert.py tests/test_gpu_modeac.py tests/test_gpu_formats.py tests/test_gpu_large.py -x -q 2>&1 | tail -15
