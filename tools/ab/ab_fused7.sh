mkdir -p gpurun_out/$1; O=gpurun_out/$1
run() { tag=$1; shift; env MGPU_LIBRARY=libmodes_gpu_exp.so MGPU_D2H_HOLD=0 "$@" timeout 300 python bench.py --no-cpu-baseline --no-extra-configs --steps 20 > $O/$tag.log 2>&1; tail -1 $O/$tag.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['stage_ms']; print('$tag', d['value'], d.get('ms_per_feed'), s['sweep'], s['slice'], s['prescreen'], 'host', s['d2h'], s['resolve_host'], s['build_host'], s['build_wait'], d['roofline']['frac'])" 2>/dev/null || tail -5 $O/$tag.log; }
for i in 1 2; do
run nomask_$i X=1
run m8_$i MGPU_CU_MASK_STRIDE=8
run m16_$i MGPU_CU_MASK_STRIDE=16
run m4_$i MGPU_CU_MASK_STRIDE=4
run m8nohold_$i MGPU_CU_MASK_STRIDE=8 MGPU_S2_HOLD=0
done
