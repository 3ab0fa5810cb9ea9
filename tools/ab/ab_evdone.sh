#!/bin/bash
# the chunk's completion as an event WITHOUT a timestamp (new) against the timed ev[3] behind every chunk (old = HEAD's library), interleaved
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q -x -k "parity or deferred or formats or large or shard or golden or cabi" 2>&1 | tail -2
p() { tail -1 $1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['stage_ms']; print('$2', d['value'], d.get('ms_per_feed'), s['sweep'], s['slice'], s['prescreen'], 'host', s['d2h'], s['resolve_host'], s['build_host'])" 2>/dev/null || tail -3 $1; }
for i in $(seq ${2:-8}); do
  timeout 300 python bench.py --no-cpu-baseline --no-extra-configs > $O/new$i.log 2>&1; p $O/new$i.log "new"
  MGPU_LIBRARY=libmodes_gpu_old.so timeout 300 python bench.py --no-cpu-baseline --no-extra-configs > $O/old$i.log 2>&1; p $O/old$i.log "old"
done 2>&1 | tee $O/evdone.txt
