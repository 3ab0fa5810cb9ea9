#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1; mkdir -p $O
cd $R
export MGPU_LIBRARY=libmodes_gpu_exp.so
p() { tail -1 $1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['stage_ms']; print('$2', d['value'], d.get('ms_per_feed'), s['sweep'], s['slice'], s['prescreen'], 'host', s['d2h'], s['resolve_host'], s['build_host'], s['build_wait'])" 2>/dev/null || tail -3 $1; }
for i in 1 2 3 4; do for v in "1 4" "0 1" "1 1" "0 4" "1 2"; do
  set -- $v
  MGPU_BUILD_PRE=$1 MGPU_BUILD_PARTS=$2 timeout 300 python bench.py --no-cpu-baseline --no-extra-configs > $O/v$1$2_$i.log 2>&1; p $O/v$1$2_$i.log "pre $1 parts/thread $2"
done; done 2>&1 | tee $O/build3.txt
