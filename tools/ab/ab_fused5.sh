mkdir -p gpurun_out/$1; O=gpurun_out/$1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_deferred.py tests/test_gpu_shard.py -x -q 2>&1 | tail -4 > $O/tests.txt; cat $O/tests.txt
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --no-cpu-baseline --no-extra-configs --steps 20 > $O/$tag.log 2>&1; tail -1 $O/$tag.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['stage_ms']; print('$tag', d['value'], d.get('ms_per_feed'), s['sweep'], s['slice'], s['prescreen'], 'host', s['d2h'], s['resolve_host'], s['build_host'], d['roofline']['frac'], d['roofline']['kernel'])" 2>/dev/null || tail -5 $O/$tag.log; }
for i in 1 2 3 4; do
run fused_$i X=1
done
