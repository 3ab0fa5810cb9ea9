# the aggregator path (one rank, RCCL, device-resident messages) against the plain path, interleaved pairs — profiles/r06_gather_vs_plain.txt
mkdir -p gpurun_out/$1; O=gpurun_out/$1; shift
p() { tail -1 $1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['stage_ms']; print('$2', d['value'], d.get('ms_per_feed'), {k: s[k] for k in ('convert','sweep','slice','prescreen','d2h','resolve_host','build_host','build_wait','sigpower')}, d.get('per_rank_host_ms'))" 2>/dev/null || tail -3 $1; }
for i in 1 2 3 4 5; do
  timeout 300 python bench.py --no-cpu-baseline --no-extra-configs "$@" > $O/plain$i.log 2>&1; p $O/plain$i.log plain
  timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2950$i bench.py --gpus 1 --exercise-gather --no-cpu-baseline --no-extra-configs "$@" > $O/gather$i.log 2>&1; p $O/gather$i.log gather
done
