#!/bin/bash
# one GPU-box session: whatever the current investigation needs, outputs under gpurun_out/<tag>/
tag=${1:-s1}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
R=$(pwd)
echo "== dev_check (product library)"
timeout 400 python tools/dev_check.py > $out/dev_check.log 2>&1; echo "rc=$?"
cat $out/dev_check.log | cut -c1-300
for blocks in 1024 768 512 256; do
  echo "== bench MGPU_SLICE_BLOCKS=$blocks"
  MGPU_SLICE_BLOCKS=$blocks timeout 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $out/bench_$blocks.log 2>&1
  tail -1 $out/bench_$blocks.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['stage_ms'])"
done
for blocks in 1536 1024 768; do
  echo "== bench MGPU_SWEEP_BLOCKS=$blocks"
  MGPU_SWEEP_BLOCKS=$blocks timeout 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $out/bench_sw$blocks.log 2>&1
  tail -1 $out/bench_sw$blocks.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['stage_ms'])"
done
