mkdir -p gpurun_out/$1; O=gpurun_out/$1
run() { tag=$1; shift; timeout 1200 python bench.py --config 5 --emulate-ranks 8 --samples 8640000000 --steps 2 --warmup 1 --no-cpu-baseline "$@" > $O/$tag.log 2>&1; tail -1 $O/$tag.log | python -c "
import sys,json
d=json.loads(sys.stdin.read()); e=d['emulated_ranks']
print('$tag', d['ms_per_step'], e['rank_critical_path_ms'], e['projected_speedup_without_communication'])
print('   ', [ (r['prepass'], r['stream_pass'], r['collect'], r['sum_blocks']) for r in e['per_rank_ms']])
" || tail -5 $O/$tag.log; }
run c1024
run c512 --chunk-buffers 512
run c256 --chunk-buffers 256
run c1024b
run c512b --chunk-buffers 512
