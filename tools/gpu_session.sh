#!/bin/bash
tag=${1:-s1}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
R=$(pwd)
cd /tmp
MASTER_ADDR=127.0.0.1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/stats_g -o bench -- python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29531 $R/bench.py --gpus 1 --steps 6 --warmup 2 --exercise-gather --no-cpu-baseline > $R/$out/g.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/stats_p -o bench -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $R/$out/p.log 2>&1
cd $R
find $out -name "*kernel_trace.csv" | head; tail -1 $out/g.log | cut -c1-200
