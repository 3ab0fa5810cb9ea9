mkdir -p gpurun_out/r06d; O=gpurun_out/r06d
timeout 900 python -m pytest tests/test_gpu_deferred.py tests/test_gpu_parity.py tests/test_cabi.py -x -q 2>&1 | tail -8 > $O/tests.txt; cat $O/tests.txt
run() { tag=$1; shift; timeout 300 python bench.py --no-cpu-baseline --no-extra-configs "$@" > $O/$tag.log 2>&1; tail -1 $O/$tag.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['value'], d['ms_per_step'], d.get('ms_per_feed'), d['stage_ms'], d['roofline']['frac'])" 2>/dev/null || tail -5 $O/$tag.log; }
for i in 1 2 3; do
run dev$i
run host$i --host-build
done
timeout 600 python bench.py --no-extra-configs > $O/full.log 2>&1; tail -1 $O/full.log | cut -c1-1500
