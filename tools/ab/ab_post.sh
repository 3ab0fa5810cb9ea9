mkdir -p gpurun_out/$1; O=gpurun_out/$1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_deferred.py tests/test_gpu_shard.py -x -q 2>&1 | tail -4 > $O/tests.txt; cat $O/tests.txt
run() { tag=$1; shift; env MGPU_LIBRARY=libmodes_gpu_exp.so "$@" timeout 300 python bench.py --no-cpu-baseline --no-extra-configs --steps 20 > $O/$tag.log 2>&1; tail -1 $O/$tag.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['stage_ms']; print('$tag', d['value'], d.get('ms_per_feed'), s['sweep'], s['slice'], s['prescreen'], 'host', s['d2h'], s['resolve_host'], s['build_host'], s['build_wait'], d['roofline']['frac'])" 2>/dev/null || tail -5 $O/$tag.log; }
for i in 1 2 3; do
run main_$i X=1
run w1_$i MGPU_WRITE_BESIDE=1
run w2_$i MGPU_WRITE_BESIDE=2
run w3_$i MGPU_WRITE_BESIDE=3
run w4_$i MGPU_WRITE_BESIDE=4
done
