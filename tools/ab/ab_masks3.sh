#!/bin/bash
# the product library with full masks as the default: UC8 plain x3, SC16Q11 --aggressive x2, the other two extras, a subset of the GPU suite
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/$1; mkdir -p $out
cd $R
export MGPU_DBG_BENCH_REPS=3
run_uc8() { echo "== uc8 [$1]"; env $1 timeout 300 python bench.py --no-cpu-baseline --no-extra-configs 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); s=d['stage_ms']; print(d['value'], d['ms_per_feed'], s['sweep'], s['slice'], s['prescreen'], 'host', s['d2h'], s['resolve_host'], s['build_host'], d['roofline']['frac'])"; }
{
for rep in 1 2 3; do run_uc8 "X=product"; done
for i in 0 1 2 0; do timeout 300 python tools/extra_reps.py $i 2>&1 | grep -v '^    {' | tail -1 | cut -c1-200; done
} 2>&1 | tee $out/masks3.txt
timeout 900 python -m pytest tests -m gpu -q -x -k "formats or modeac or deferred or parity or gather or fanin" 2>&1 | tail -3
