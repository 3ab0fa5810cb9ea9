mkdir -p gpurun_out/$1; O=gpurun_out/$1
MGPU_LIBRARY=libmodes_gpu_exp.so MGPU_SWEEP_FUSED=3 timeout 900 python -m pytest tests/test_gpu_formats.py tests/test_gpu_parity.py -x -q 2>&1 | tail -6 | cut -c1-400
for i in 1 2; do
for f in 1 3; do
  echo "MGPU_SWEEP_FUSED=$f"; MGPU_LIBRARY=libmodes_gpu_exp.so MGPU_SWEEP_FUSED=$f MGPU_DBG_BENCH_REPS=3 timeout 500 python tools/extra_reps.py 0 2>&1 | tail -4 | tee -a $O/reps_f$f.txt
done
done
