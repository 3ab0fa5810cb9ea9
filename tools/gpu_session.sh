#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/s60
O=gpurun_out/s60
timeout 600 python bench.py --steps 10 --warmup 2 --no-extra-configs 2>$O/bench.err | tail -1 > $O/bench.json; cut -c1-1300 $O/bench.json
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_device_walk.py tests/test_gpu_formats.py -m gpu -q -x 2>&1 | tail -3
