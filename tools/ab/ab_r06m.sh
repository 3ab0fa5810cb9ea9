mkdir -p gpurun_out/r06m; O=gpurun_out/r06m
for i in 1 2 3 4 5 6 7 8 9 10; do timeout 300 python bench.py --no-cpu-baseline --no-extra-configs > $O/p$i.log 2>&1; tail -1 $O/p$i.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['stage_ms']; print('plain', d['value'], d['ms_per_feed'], s['resolve_host'], s['build_host'], s['d2h'])"; done
bash tools/host_4rank.sh r06m4
