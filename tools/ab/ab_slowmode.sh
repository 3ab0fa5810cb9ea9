#!/bin/bash
# hunting the processes whose host stages run slow: N plain benchmark processes with MGPU_DBG_BENCH_HOST=1
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1; mkdir -p $O
cd $R
for i in $(seq ${2:-16}); do
  MGPU_DBG_BENCH_HOST=1 timeout 300 python bench.py --no-cpu-baseline --no-extra-configs > $O/p$i.log 2> $O/p$i.err
  tail -1 $O/p$i.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['stage_ms']; print('process $i', d['value'], d.get('ms_per_feed'), 'host', s['d2h'], s['resolve_host'], s['build_host'])"
  grep '^dbg host' $O/p$i.err | cut -c1-1500
done 2>&1 | tee $O/slowmode.txt
