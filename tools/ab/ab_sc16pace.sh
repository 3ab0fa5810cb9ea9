#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/$1; mkdir -p $out
cd $R
export MGPU_LIBRARY=libmodes_gpu_exp.so MGPU_DBG_BENCH_REPS=3
for rep in 1 2; do for v in "X=1" "MGPU_SWEEP_PACE=0" "MGPU_SWEEP_PACE=800"; do
  echo "== [$v] (rep $rep)"; env $v timeout 300 python tools/extra_reps.py 0 2>&1 | tail -4 | cut -c1-200
done; done 2>&1 | tee $out/sc16pace.txt
