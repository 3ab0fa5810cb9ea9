#!/bin/bash
tag=${1:-s1}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
( time timeout 900 python bench.py > $out/bench_default.log 2>$out/bench_default.err ) 2>&1 | grep real
tail -1 $out/bench_default.log | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['stage_ms']); print(d['roofline']['frac'], d['roofline']['traffic'], d['kernels']['k_slice']['traffic'])
for k,v in d.get('configs',{}).items(): print(k, v['msamples_s'], v['ms_per_segment'], v['us_per_launch'])"
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 600 > $out/pytest_gpu.log 2>&1; tail -5 $out/pytest_gpu.log
