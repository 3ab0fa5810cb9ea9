mkdir -p gpurun_out/$1; O=gpurun_out/$1
p() { tail -1 $1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['stage_ms']; print('$2', d['value'], d.get('ms_per_feed'), {k: s[k] for k in ('convert','sweep','slice','prescreen','d2h','resolve_host','build_host','build_wait','sigpower')})" 2>/dev/null || tail -3 $1; }
g() { tag=$1; shift; env "$@" timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --exercise-gather --no-cpu-baseline --no-extra-configs > $O/$tag.log 2>&1; p $O/$tag.log $tag; }
for i in 1 2 3 4 5; do
  timeout 300 python bench.py --no-cpu-baseline --no-extra-configs > $O/plain$i.log 2>&1; p $O/plain$i.log plain
  g gather$i X=1
done
