#!/bin/bash
# which side stream needs the queue of its own: both (3, the product), only the second stream (1), only the record copies (2)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1; mkdir -p $O
cd $R
export MGPU_LIBRARY=libmodes_gpu_exp.so
p() { tail -1 $1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['stage_ms']; print('$2', d['value'], d.get('ms_per_feed'), s['sweep'], s['slice'], s['prescreen'], 'host', s['d2h'], s['resolve_host'], s['build_host'])" 2>/dev/null || tail -3 $1; }
for i in 1 2 3 4; do for w in 3 1 2; do
  env MGPU_OWN_QUEUES=$w timeout 300 python bench.py --no-cpu-baseline --no-extra-configs > $O/q${w}_$i.log 2>&1; p $O/q${w}_$i.log "own queues $w"
done; done 2>&1 | tee $O/ownq.txt
