mkdir -p gpurun_out/r06c; O=$GRAFT_REPO_ROOT/gpurun_out/r06c; R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
for mode in side main; do
  if [ $mode = side ]; then E="MGPU_CONV_SIDE=1 MGPU_SLICE_BLOCKS=768"; else E="X=1"; fi
  env MGPU_LIBRARY=libmodes_gpu_exp.so $E timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/tr_$mode -o t -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra-configs --event-bracket-us 4.0 > $O/tr_$mode.log 2>&1
  f=$(find $O/tr_$mode -name "*kernel_trace.csv" | head -1)
  python $R/tools/trace_compact.py $f $O/trace_$mode.txt
  rm -rf $O/tr_$mode
  tail -1 $O/tr_$mode.log | cut -c1-400
done
