mkdir -p gpurun_out/$1; O=gpurun_out/$1
run() { tag=$1; lib=$2; shift; shift; env MGPU_LIBRARY=$lib "$@" timeout 300 python bench.py --no-cpu-baseline --no-extra-configs --steps 10 > $O/$tag.log 2>&1; tail -1 $O/$tag.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['stage_ms']; print('$tag', d['value'], d.get('ms_per_feed'), s['convert'], s['sweep'], s['slice'], s['prescreen'], 'host', s['d2h'], s['resolve_host'], s['build_host'])" 2>/dev/null || tail -5 $O/$tag.log; }
for i in 1 2 3; do
run plain_$i libmodes_gpu.so
run nt_$i libmodes_gpu_nt.so
done
