mkdir -p gpurun_out/r06t; O=gpurun_out/r06t
run() { tag=$1; lib=$2; shift; shift; MGPU_LIBRARY=$lib timeout 300 python bench.py --no-cpu-baseline --no-extra-configs --steps 10 "$@" > $O/$tag.log 2>&1; tail -1 $O/$tag.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['stage_ms']; print('$tag', d['value'], d.get('ms_per_feed'), s['convert'], s['sweep'], s['slice'], s['prescreen'])" 2>/dev/null || tail -5 $O/$tag.log; }
for i in 1 2; do
run d4_$i libmodes_gpu.so
run d2_$i libmodes_gpu_cv2.so
run d8_$i libmodes_gpu_cv8.so
done
