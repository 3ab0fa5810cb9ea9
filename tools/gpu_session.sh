#!/bin/bash
tag=${1:-s1}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
MGPU_LIBRARY=libmodes_gpu_tm.so MGPU_DEBUG_PRINT=1 timeout 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | grep "dbg: k_sweep" | tail -12
