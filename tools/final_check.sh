#!/bin/bash
# the whole GPU suite, smoke(), and the default benchmark line on the committed code: tools/final_check.sh <tag>
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/${1:-final}
mkdir -p $out
cd $R
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -6 > $out/gpu_suite_tail.txt; cat $out/gpu_suite_tail.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 600 python bench.py > $out/bench_default.log 2> $out/bench_default.err; tail -1 $out/bench_default.log | cut -c1-1500; tail -3 $out/bench_default.err
