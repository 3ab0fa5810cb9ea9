#!/bin/bash
# one GPU-box session: whatever the current investigation needs, outputs under gpurun_out/<tag>/
tag=${1:-s1}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
R=$(pwd)
echo "== new tests"
timeout 600 python -m pytest tests/test_gpu_deferred.py tests/test_gpu_modeac.py tests/test_gpu_kernel_generations.py -x -q --timeout 300 > $out/pytest_new.log 2>&1; tail -15 $out/pytest_new.log
echo "== bench"
timeout 400 python bench.py --steps 10 --warmup 3 > $out/bench.log 2>&1; tail -2 $out/bench.log | cut -c1-2500
echo "== full gpu suite"
timeout 1200 python -m pytest tests -m gpu -q -x --timeout 600 > $out/pytest_gpu.log 2>&1; tail -8 $out/pytest_gpu.log
