#!/bin/bash
# the fan-in CLI with 4 / 8 streams: the product library against the experiments library with ordinary side streams
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/$1; mkdir -p $out
cd $R
mkdir -p /tmp/explib && cp readsb_amd/csrc/libmodes_gpu_exp.so /tmp/explib/libmodes_gpu.so
readelf -d readsb_amd/host/readsb_gpu_fanin | grep -i 'rpath\|runpath'
{
echo "== product"; timeout 600 python tools/bench_fanin.py --seconds 300 --streams 4,8 2>&1 | tail -2
echo "== experiments library, default"; LD_LIBRARY_PATH=/tmp/explib timeout 600 python tools/bench_fanin.py --seconds 300 --streams 4,8 2>&1 | tail -2
echo "== experiments library, ordinary side streams"; LD_LIBRARY_PATH=/tmp/explib MGPU_CU_MASK_STRIDE=0 timeout 600 python tools/bench_fanin.py --seconds 300 --streams 4,8 2>&1 | tail -2
echo "== experiments library, ordinary side streams, GPU_MAX_HW_QUEUES=8"; GPU_MAX_HW_QUEUES=8 LD_LIBRARY_PATH=/tmp/explib MGPU_CU_MASK_STRIDE=0 timeout 600 python tools/bench_fanin.py --seconds 300 --streams 4,8 2>&1 | tail -2
} 2>&1 | tee $out/fanin.txt
