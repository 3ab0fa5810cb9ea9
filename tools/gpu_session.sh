#!/bin/bash
tag=${1:-s1}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
run() { name=$1; shift; echo "== $name"; env "$@" timeout 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $out/bench_$name.log 2>&1; tail -1 $out/bench_$name.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['stage_ms'])"; }
run plain1 X=1
run plain2 X=1
timeout 300 python tools/dev_check.py uc8_fix_2s uc8_dense uc8_fix_30s | cut -c1-60
for i in 1 2; do
echo "== exercise gather $i"
MASTER_ADDR=127.0.0.1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2951$i bench.py --gpus 1 --steps 20 --warmup 3 --exercise-gather --no-cpu-baseline > $out/bench_gather$i.log 2>&1; tail -1 $out/bench_gather$i.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['stage_ms'])"
done
