#!/bin/bash
tag=${1:-s1}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
run() { name=$1; shift; echo "== $name"; env "$@" timeout 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $out/bench_$name.log 2>&1; tail -1 $out/bench_$name.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['stage_ms'])"; }
run one_l3 MGPU_ONE_L3=1
run two_l3_8_6 X=1
run two_l3_8_4 MGPU_BUILD_THREADS=4
run two_l3_6_6 MGPU_WALK_THREADS=6
run two_l3_8_6_b X=1
run two_l3_7_6 MGPU_WALK_THREADS=7
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_deferred.py tests/test_gpu_fanin.py -x -q --timeout 300 2>&1 | tail -3
