#!/bin/bash
# the builder stage with the statistics' divisions in the team's ranges and four ranges per thread: parity tests, then old (HEAD's library built as a variant) against new, interleaved
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -q -x -k "parity or deferred or formats or large or modeac or shard or golden or pipeline_chain or dropin" 2>&1 | tail -3
p() { tail -1 $1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['stage_ms']; print('$2', d['value'], d.get('ms_per_feed'), 'host', s['d2h'], s['resolve_host'], s['build_host'], s['build_wait'])" 2>/dev/null || tail -3 $1; }
for i in 1 2 3 4; do
  timeout 300 python bench.py --no-cpu-baseline --no-extra-configs > $O/new$i.log 2>&1; p $O/new$i.log "new"
  MGPU_LIBRARY=libmodes_gpu_old.so timeout 300 python bench.py --no-cpu-baseline --no-extra-configs > $O/old$i.log 2>&1; p $O/old$i.log "old"
done 2>&1 | tee $O/build.txt
MGPU_LIBRARY=libmodes_gpu_exp.so MGPU_DEBUG_PRINT=1 timeout 300 python bench.py --no-cpu-baseline --no-extra-configs --steps 2 --warmup 1 2> $O/dbgprint.err > /dev/null; grep "dbg: build tasks" $O/dbgprint.err | tail -3 | cut -c1-700
