#!/bin/bash
# the main stream with a hardware queue of its own (the product's default now) against an ordinary one: plain and aggregator path, interleaved
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1; mkdir -p $O
cd $R
export MGPU_LIBRARY=libmodes_gpu_exp.so
p() { tail -1 $1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['stage_ms']; print('$2', d['value'], d.get('ms_per_feed'), s['sweep'], s['slice'], s['prescreen'], 'host', s['d2h'], s['resolve_host'], s['build_host'])" 2>/dev/null || tail -3 $1; }
g() { tag=$1; shift; env "$@" timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --exercise-gather --no-cpu-baseline --no-extra-configs > $O/$tag.log 2>&1; p $O/$tag.log "$tag"; }
for i in 1 2 3; do
  env MGPU_MAIN_OWN_QUEUE=1 timeout 300 python bench.py --no-cpu-baseline --no-extra-configs > $O/plain_own$i.log 2>&1; p $O/plain_own$i.log "plain own"
  env MGPU_MAIN_OWN_QUEUE=0 timeout 300 python bench.py --no-cpu-baseline --no-extra-configs > $O/plain_ord$i.log 2>&1; p $O/plain_ord$i.log "plain ordinary"
  g "gather_own$i" MGPU_MAIN_OWN_QUEUE=1
  g "gather_ord$i" MGPU_MAIN_OWN_QUEUE=0
done 2>&1 | tee $O/mainq.txt
