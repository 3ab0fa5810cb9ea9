mkdir -p gpurun_out/r06g; O=gpurun_out/r06g
timeout 1200 python -m pytest tests/test_gpu_gate.py tests/test_gpu_beast.py -x -q 2>&1 | tail -15 > $O/tests.txt; cat $O/tests.txt
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/gate -o g -- python $GRAFT_REPO_ROOT/tools/bench_gate.py > $GRAFT_REPO_ROOT/$O/bench_gate.log 2>&1
cd $GRAFT_REPO_ROOT; tail -3 $O/bench_gate.log | cut -c1-400; f=$(find $O/gate -name "*kernel_stats.csv" | head -1); cut -d, -f1-6 $f | head -14 | cut -c1-150
