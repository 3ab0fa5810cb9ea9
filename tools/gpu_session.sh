#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/s66
O=gpurun_out/s66
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -4 | tee $O/pytest.log
timeout 600 python tools/bench_fanin.py --seconds 60 --streams 1,2,4,8 2>&1 | tail -6 | tee $O/fanin.txt
MGPU_WALK_THREADS=4 MGPU_BUILD_THREADS=3 timeout 300 python tools/bench_fanin.py --seconds 60 --streams 8 2>&1 | tail -2 | tee -a $O/fanin.txt
timeout 300 python tools/bench_formats.py 2>&1 | tail -8 | tee $O/formats.txt
