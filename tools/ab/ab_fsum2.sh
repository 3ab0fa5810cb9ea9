#!/bin/bash
# SC16 formats: the float-sum chains of consecutive chunks on two streams in turn (the product now) against one (MGPU_FSUM_STREAMS=1)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/$1; mkdir -p $out
cd $R
timeout 900 python -m pytest tests -m gpu -q -x -k "sc16 or SC16 or formats or modeac or convert or fsum or generations or deferred" 2>&1 | tail -3
export MGPU_LIBRARY=libmodes_gpu_exp.so MGPU_DBG_BENCH_REPS=3
for rep in 1 2 3; do for v in 2 1; do
  echo "== fsum streams $v (rep $rep)"; MGPU_FSUM_STREAMS=$v timeout 300 python tools/extra_reps.py 0 2>&1 | tail -4 | cut -c1-260
done; done 2>&1 | tee $out/fsum2.txt
