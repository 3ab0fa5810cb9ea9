#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/$1; mkdir -p $out
cd $R
mkdir -p /tmp/explib && cp readsb_amd/csrc/libmodes_gpu_exp.so /tmp/explib/libmodes_gpu.so
export LD_LIBRARY_PATH=/tmp/explib
{
echo "== default (4 walkers + 3 builders per context)"; timeout 600 python tools/bench_fanin.py --seconds 300 --streams 8 2>&1 | tail -1
echo "== 2 + 2"; MGPU_WALK_THREADS=2 MGPU_BUILD_THREADS=2 timeout 600 python tools/bench_fanin.py --seconds 300 --streams 8 2>&1 | tail -1
echo "== 1 + 1"; MGPU_WALK_THREADS=1 MGPU_BUILD_THREADS=1 timeout 600 python tools/bench_fanin.py --seconds 300 --streams 8 2>&1 | tail -1
echo "== 2 + 2, chunk 2048"; FANIN_CHUNK=2048 MGPU_WALK_THREADS=2 MGPU_BUILD_THREADS=2 timeout 600 python tools/bench_fanin.py --seconds 300 --streams 8 2>&1 | tail -1
echo "== default, chunk 2048"; FANIN_CHUNK=2048 timeout 600 python tools/bench_fanin.py --seconds 300 --streams 4,8 2>&1 | tail -2
} 2>&1 | tee $out/fanin5.txt
