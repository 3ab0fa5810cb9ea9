#!/bin/bash
tag=${1:-s1}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
export MGPU_COPY_AFTER_SWEEP=0
p() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['stage_ms'])"; }
R=$(pwd)
try() { name=$1; shift; echo "== $name"; env "$@" timeout 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | p
cd /tmp; env "$@" timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/stats_$name -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/$out/rp_$name.log 2>&1; cd $R
grep -c copyBuffer $out/stats_$name/bench_kernel_trace.csv; }
try base X=1
try sdma HSA_ENABLE_SDMA=1
try blit0 GPU_FORCE_BLIT_COPY_SIZE=0
try sdma_blit0 HSA_ENABLE_SDMA=1 GPU_FORCE_BLIT_COPY_SIZE=0
printenv | grep -i "HSA_\|GPU_\|ROC" | head
