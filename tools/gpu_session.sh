#!/bin/bash
tag=${1:-s1}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
R=$(pwd)
for rep in 1 2 3; do
for cb in 0 8 16 32; do
echo "== rep $rep bench MGPU_COPY_BLOCKS=$cb"
MGPU_COPY_BLOCKS=$cb timeout 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $out/bench_cb${cb}_$rep.log 2>&1; tail -1 $out/bench_cb${cb}_$rep.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['stage_ms'])"
done
done
