#!/bin/bash
# the product library with the float-sum chain's mask as the default: SC16Q11 --aggressive, plain SC16, SC16 + Mode A/C parity tests, and the bench's extras
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/$1; mkdir -p $out
cd $R
timeout 900 python -m pytest tests -m gpu -q -x -k "sc16 or SC16 or formats or modeac or convert or fsum" 2>&1 | tail -3
export MGPU_DBG_BENCH_REPS=3
for rep in 1 2 3; do echo "== product (rep $rep)"; timeout 300 python tools/extra_reps.py 0 2>&1 | tail -4 | cut -c1-300; done 2>&1 | tee $out/fsum_mask_product.txt
MGPU_LIBRARY=libmodes_gpu_exp.so MGPU_FSUM_CU_STRIDE=0 timeout 300 python tools/extra_reps.py 0 2>&1 | tail -4 | cut -c1-300 | tee -a $out/fsum_mask_product.txt
