mkdir -p gpurun_out/$1; O=gpurun_out/$1
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "long_chunk" 2>&1 | tail -40 > $O/tests.txt; cat $O/tests.txt | cut -c1-300
run() { tag=$1; shift; timeout 300 python bench.py --no-cpu-baseline --no-extra-configs --steps 20 "$@" > $O/$tag.log 2>&1; tail -1 $O/$tag.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['stage_ms']; print('$tag', d['value'], d.get('ms_per_feed'), s['sweep'], s['slice'], s['prescreen'], 'host', s['d2h'], s['resolve_host'], s['build_host'], s['build_wait'], d['roofline']['frac'])" 2>/dev/null || tail -5 $O/$tag.log; }
for i in 1 2; do
run c1024a3_$i --ahead 3
run c2048a2_$i --chunk-buffers 2048 --ahead 2
run c2048a3_$i --chunk-buffers 2048 --ahead 3
run c4096a3_$i --chunk-buffers 4096 --ahead 3
done
