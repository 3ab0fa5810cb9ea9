#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/$1; mkdir -p $out
cd $R
{
for c in 128 256 1024; do echo "== chunk $c"; FANIN_CHUNK=$c timeout 600 python tools/bench_fanin.py --seconds 300 --streams 4,8 2>&1 | tail -2; done
echo "== chunk 512, 5 6 7 streams"; timeout 600 python tools/bench_fanin.py --seconds 300 --streams 5,6,7 2>&1 | tail -3
} 2>&1 | tee $out/fanin3.txt
