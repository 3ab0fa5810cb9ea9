mkdir -p gpurun_out/r06f; O=gpurun_out/r06f
run() { tag=$1; lib=$2; shift; shift; MGPU_LIBRARY=$lib timeout 300 python bench.py --no-cpu-baseline --no-extra-configs "$@" > $O/$tag.log 2>&1; tail -1 $O/$tag.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['stage_ms']; g=s['convert']+s['sweep']+s['slice']+s['prescreen']; print('$tag', d['value'], d.get('ms_per_feed'), 'gpu-sum', round(g,3), s, d['roofline']['frac'])" 2>/dev/null || tail -5 $O/$tag.log; }
for i in 1 2; do
run s4_$i libmodes_gpu.so
run s8_$i libmodes_gpu_s8.so
run s8a2_$i libmodes_gpu_s8.so --ahead 2
run s12_$i libmodes_gpu_s12.so
run s12a2_$i libmodes_gpu_s12.so --ahead 2
run s10_$i libmodes_gpu_s10.so
done
