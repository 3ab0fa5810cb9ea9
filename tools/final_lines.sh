mkdir -p gpurun_out/$1; O=gpurun_out/$1
timeout 900 python bench.py > $O/bench_default.log 2>&1; tail -1 $O/bench_default.log | cut -c1-3500
for i in 1 2 3; do timeout 300 python bench.py --no-cpu-baseline --no-extra-configs > $O/plain_$i.log 2>&1; tail -1 $O/plain_$i.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['stage_ms']; print('plain', d['value'], d['ms_per_step'], d.get('ms_per_feed'), s['sweep'], s['slice'], s['prescreen'], 'host', s['d2h'], s['resolve_host'], s['build_host'], d['roofline']['frac'], d['roofline']['traffic'])"; done
