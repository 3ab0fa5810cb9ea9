#!/bin/bash
# is it the queue?  full masks (all CUs) against the strided ones, the main stream with a queue of its own, GPU_MAX_HW_QUEUES=8 with plain streams
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/$1; mkdir -p $out
cd $R
export MGPU_LIBRARY=libmodes_gpu_exp.so MGPU_DBG_BENCH_REPS=3
run_sc16() { echo "== sc16 [$1]"; env $1 timeout 300 python tools/extra_reps.py 0 2>&1 | grep 'configs\[2\]' | cut -c1-200; }
run_uc8() { echo "== uc8 [$1]"; env $1 timeout 300 python bench.py --no-cpu-baseline --no-extra-configs 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); s=d['stage_ms']; print(d['value'], d['ms_per_feed'], s['sweep'], s['slice'], s['prescreen'], 'host', s['d2h'], s['resolve_host'], s['build_host'], d['roofline']['frac'])"; }
for rep in 1 2 3; do
  for v in "X=1" "MGPU_CU_MASK_PERXCC=32,0" "MGPU_CU_MASK_PERXCC=32,0 MGPU_MAIN_OWN_QUEUE=1" "MGPU_MAIN_OWN_QUEUE=1" "GPU_MAX_HW_QUEUES=8 MGPU_CU_MASK_STRIDE=0 MGPU_S2_PRIORITY=0" "GPU_MAX_HW_QUEUES=8 MGPU_CU_MASK_STRIDE=0"; do run_uc8 "$v"; done
done 2>&1 | tee $out/masks2.txt
for rep in 1 2; do
  for v in "GPU_MAX_HW_QUEUES=8 MGPU_FSUM_CU_STRIDE=0 MGPU_FSUM_PRIORITY=-1" "GPU_MAX_HW_QUEUES=8 MGPU_FSUM_CU_STRIDE=0 MGPU_FSUM_PRIORITY=0" "MGPU_FSUM_CU_PERXCC=32,0 MGPU_MAIN_OWN_QUEUE=1"; do run_sc16 "$v"; done
done 2>&1 | tee -a $out/masks2.txt
