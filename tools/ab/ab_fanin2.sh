#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/$1; mkdir -p $out
cd $R
mkdir -p /tmp/explib && cp readsb_amd/csrc/libmodes_gpu_exp.so /tmp/explib/libmodes_gpu.so
export LD_LIBRARY_PATH=/tmp/explib
{
echo "== default"; timeout 600 python tools/bench_fanin.py --seconds 300 --streams 4,8 2>&1 | tail -2
echo "== main stream own queue"; MGPU_MAIN_OWN_QUEUE=1 timeout 600 python tools/bench_fanin.py --seconds 300 --streams 4,8 2>&1 | tail -2
echo "== main + upload stream own queue"; MGPU_MAIN_OWN_QUEUE=2 timeout 600 python tools/bench_fanin.py --seconds 300 --streams 4,8 2>&1 | tail -2
echo "== unpinned"; MGPU_NO_AFFINITY=1 timeout 600 python tools/bench_fanin.py --seconds 300 --streams 4,8 2>&1 | tail -2
echo "== GPU_MAX_HW_QUEUES=24"; GPU_MAX_HW_QUEUES=24 timeout 600 python tools/bench_fanin.py --seconds 300 --streams 4,8 2>&1 | tail -2
} 2>&1 | tee $out/fanin2.txt
