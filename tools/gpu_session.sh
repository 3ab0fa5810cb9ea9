#!/bin/bash
tag=${1:-s1}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
p() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['stage_ms']['resolve_host'], d['stage_ms']['build_host'])"; }
for i in 1 2 3; do
echo "== gather device msgs $i"; MASTER_ADDR=127.0.0.1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2954$i bench.py --gpus 1 --steps 20 --warmup 3 --exercise-gather --no-cpu-baseline 2>/dev/null | tail -1 | p
echo "== gather host msgs $i"; MGPU_DBG_HOST_MESSAGES=1 MASTER_ADDR=127.0.0.1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2955$i bench.py --gpus 1 --steps 20 --warmup 3 --exercise-gather --no-cpu-baseline 2>/dev/null | tail -1 | p
done
echo "== plain"; timeout 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | p
