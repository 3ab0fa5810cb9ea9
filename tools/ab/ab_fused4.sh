mkdir -p gpurun_out/$1; O=gpurun_out/$1
run() { tag=$1; shift; env MGPU_LIBRARY=libmodes_gpu_exp.so "$@" timeout 300 python bench.py --no-cpu-baseline --no-extra-configs --steps 20 > $O/$tag.log 2>&1; tail -1 $O/$tag.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['stage_ms']; print('$tag', d['value'], d.get('ms_per_feed'), s['sweep'], s['slice'], s['prescreen'], 'host', s['d2h'], s['resolve_host'], s['build_host'])" 2>/dev/null || tail -5 $O/$tag.log; }
for i in 1 2 3 4 5 6; do
run hold_$i X=1
run nohold_$i MGPU_S2_HOLD=0
done
