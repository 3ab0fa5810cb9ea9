#!/bin/bash
# the first benchmark processes on a fresh box: default warm-up (5 steps = 0.28 s) against a long one
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1; mkdir -p $O
cd $R
p() { tail -1 $1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['stage_ms']; print('$2', d['value'], d.get('ms_per_feed'), s['sweep'], s['slice'], s['prescreen'], 'host', s['d2h'], s['resolve_host'], s['build_host'])" 2>/dev/null || tail -3 $1; }
for i in 1 2 3 4 5 6; do
  timeout 300 python bench.py --no-cpu-baseline --no-extra-configs --warmup $2 > $O/w$2_$i.log 2>&1; p $O/w$2_$i.log "warmup $2 process $i"
done 2>&1 | tee $O/firstproc_w$2.txt
