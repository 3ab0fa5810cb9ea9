#!/bin/bash
# What the side streams' "CU masks" really do (tools/micro/cu_mask_map.hip: a mask whose bits all fall to one XCC leaves the other
# XCCs unrestricted): priorities, strided masks and true per-XCC confinement, for the SC16 float-sum chain and for the UC8 side streams.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/$1; mkdir -p $out
cd $R
export MGPU_LIBRARY=libmodes_gpu_exp.so MGPU_DBG_BENCH_REPS=3
run_sc16() { echo "== sc16 [$1]"; env $1 timeout 300 python tools/extra_reps.py 0 2>&1 | grep 'configs\[2\]' | cut -c1-200; }
run_uc8() { echo "== uc8 [$1]"; env $1 timeout 300 python bench.py --no-cpu-baseline --no-extra-configs 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); s=d['stage_ms']; print(d['value'], d['ms_per_feed'], s['sweep'], s['slice'], s['prescreen'], 'host', s['d2h'], s['resolve_host'], s['build_host'], d['roofline']['frac'])"; }
for rep in 1 2; do
  for v in "X=1" "MGPU_FSUM_CU_STRIDE=0 MGPU_FSUM_PRIORITY=-1" "MGPU_FSUM_CU_STRIDE=0 MGPU_FSUM_PRIORITY=0" "MGPU_FSUM_CU_STRIDE=0 MGPU_FSUM_PRIORITY=1" "MGPU_FSUM_CU_PERXCC=32,0" "MGPU_FSUM_CU_PERXCC=16,0" "MGPU_FSUM_CU_PERXCC=8,0" "MGPU_FSUM_CU_PERXCC=8,8" "MGPU_FSUM_CU_PERXCC=4,4"; do run_sc16 "$v"; done
  for v in "X=1" "MGPU_CU_MASK_STRIDE=0" "MGPU_CU_MASK_STRIDE=0 MGPU_S2_PRIORITY=0" "MGPU_CU_MASK_STRIDE=0 MGPU_S2_PRIORITY=1" "MGPU_CU_MASK_PERXCC=32,0" "MGPU_CU_MASK_PERXCC=8,0" "MGPU_CU_MASK_PERXCC=4,0" "MGPU_CU_MASK_PERXCC=2,0"; do run_uc8 "$v"; done
done 2>&1 | tee $out/masks.txt
