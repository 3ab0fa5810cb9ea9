mkdir -p gpurun_out/$1; O=gpurun_out/$1
MGPU_LIBRARY=libmodes_gpu_sb4.so timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_formats.py -x -q 2>&1 | tail -3
run() { tag=$1; lib=$2; MGPU_LIBRARY=$lib timeout 300 python bench.py --no-cpu-baseline --no-extra-configs --steps 20 > $O/$tag.log 2>&1; tail -1 $O/$tag.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['stage_ms']; print('$tag', d['value'], d.get('ms_per_feed'), s['sweep'], s['slice'], s['prescreen'], 'host', s['d2h'], s['resolve_host'], s['build_host'], s['build_wait'], d['roofline']['frac'])" 2>/dev/null || tail -5 $O/$tag.log; }
for i in 1 2 3; do
run main_$i libmodes_gpu.so
run sb2_$i libmodes_gpu_sb2.so
run sb4_$i libmodes_gpu_sb4.so
run sb8_$i libmodes_gpu_sb8.so
done
