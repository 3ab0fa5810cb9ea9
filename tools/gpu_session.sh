#!/bin/bash
# one GPU-box session: whatever the current investigation needs, outputs under gpurun_out/<tag>/
tag=${1:-s1}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
( cd tools/micro && timeout 120 ./valu_issue ../../$out/valu_issue.json > ../../$out/valu_issue.txt 2>&1 )
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 > $out/pytest_gpu.log 2>&1
tail -5 $out/pytest_gpu.log
timeout 300 python bench.py --steps 10 --warmup 3 > $out/bench.log 2>&1
tail -1 $out/bench.log | cut -c1-600
cat $out/valu_issue.txt
