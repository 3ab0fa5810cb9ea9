#!/bin/bash
# SC16Q11 --aggressive (bench.py's first extra configuration): the float-sum chain's stream unmasked against every k-th CU
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/$1; mkdir -p $out
cd $R
export MGPU_LIBRARY=libmodes_gpu_exp.so MGPU_DBG_BENCH_REPS=3
for rep in 1 2; do
for v in none 2,1 4,1 4,2 8,4 3,1; do
  if [ $v = none ]; then unset MGPU_FSUM_CU_STRIDE; else export MGPU_FSUM_CU_STRIDE=$v; fi
  echo "== fsum mask $v (rep $rep)"; timeout 300 python tools/extra_reps.py 0 2>&1 | tail -4 | cut -c1-400
done; done 2>&1 | tee $out/fsum_mask.txt
