#!/bin/bash
tag=${1:-s1}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
echo "== dev_check fused"
timeout 400 python tools/dev_check.py > $out/dev_check.log 2>&1; echo "rc=$?"; grep -c "^OK" $out/dev_check.log
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_formats.py tests/test_gpu_large.py -x -q --timeout 300 > $out/pytest_sel.log 2>&1; tail -2 $out/pytest_sel.log
run() { name=$1; shift; echo "== $name"; env "$@" timeout 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $out/bench_$name.log 2>&1; tail -1 $out/bench_$name.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['stage_ms'])"; }
run separate MGPU_FUSED_CONVERT=0
run fused X=1
run separate2 MGPU_FUSED_CONVERT=0
run fused2 X=1
