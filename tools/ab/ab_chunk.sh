mkdir -p gpurun_out/$1; O=gpurun_out/$1
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "long_chunk or config2 or dense" 2>&1 | tail -4 > $O/tests.txt; cat $O/tests.txt
run() { tag=$1; shift; timeout 300 python bench.py --no-cpu-baseline --no-extra-configs --steps 20 "$@" > $O/$tag.log 2>&1; tail -1 $O/$tag.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['stage_ms']; print('$tag', d['value'], d.get('ms_per_feed'), s['sweep'], s['slice'], s['prescreen'], 'host', s['d2h'], s['resolve_host'], s['build_host'], s['build_wait'], d['roofline']['frac'])" 2>/dev/null || tail -5 $O/$tag.log; }
for i in 1 2 3; do
run c1024_$i
run c2048_$i --chunk-buffers 2048
run c4096_$i --chunk-buffers 4096
done
