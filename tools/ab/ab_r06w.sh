mkdir -p gpurun_out/r06w; O=$GRAFT_REPO_ROOT/gpurun_out/r06w; R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr -o t -- python $R/bench.py --steps 2 --warmup 1 --loops 8 --no-cpu-baseline --no-extra-configs --event-bracket-us 4.0 > $O/tr.log 2>&1
f=$(find $O/tr -name "*kernel_trace.csv" | head -1); python $R/tools/trace_compact.py $f $O/trace_fused.txt
cp $(find $O/tr -name "*kernel_stats.csv" | head -1) $O/stats_fused.csv; rm -rf $O/tr
cut -d, -f1-4 $O/stats_fused.csv | head -12 | cut -c1-120
