#!/bin/bash
# the host walk with more buffer ranges per round than walk threads (taken by ticket): the cost when nothing disturbs, the gain when a
# walker core's SMT sibling is busy (a spinning process pinned there)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1; mkdir -p $O
cd $R
export MGPU_LIBRARY=libmodes_gpu_exp.so
p() { tail -1 $1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['stage_ms']; print('$2', d['value'], d.get('ms_per_feed'), 'host', s['d2h'], s['resolve_host'], s['build_host'])" 2>/dev/null || tail -3 $1; }
# the pinned CPUs of this box
MGPU_DBG_BENCH_HOST=1 timeout 300 python bench.py --no-cpu-baseline --no-extra-configs --steps 2 --warmup 1 > $O/probe.log 2> $O/probe.err
cpu=$(grep '^dbg host' $O/probe.err | python -c "import sys,json; d=json.loads(sys.stdin.read()[9:]); print(d['pinned_cpus'][3])")
sib=$(cat /sys/devices/system/cpu/cpu$cpu/topology/thread_siblings_list | tr ',-' '\n\n' | grep -v "^$cpu\$" | head -1)
echo "walker cpu $cpu, its sibling $sib"
{
for i in 1 2 3; do for k in 8 12 16 24; do
  env MGPU_WALK_RANGES=$k timeout 300 python bench.py --no-cpu-baseline --no-extra-configs > $O/quiet_${k}_$i.log 2>&1; p $O/quiet_${k}_$i.log "quiet ranges $k"
done; done
taskset -c $sib python -c "
while True: pass" &
spin=$!
sleep 0.5
for i in 1 2; do for k in 8 12 16 24; do
  env MGPU_WALK_RANGES=$k timeout 300 python bench.py --no-cpu-baseline --no-extra-configs > $O/busy_${k}_$i.log 2>&1; p $O/busy_${k}_$i.log "sibling busy, ranges $k"
done; done
kill $spin
} 2>&1 | tee $O/ranges.txt
