mkdir -p gpurun_out/$1; O=gpurun_out/$1
run() { MGPU_LIBRARY=$2 timeout 1200 python bench.py --config 5 --emulate-ranks 8 --samples 8640000000 --steps 2 --warmup 1 --no-cpu-baseline > $O/$1.log 2>&1; tail -1 $O/$1.log | python -c "
import sys,json
d=json.loads(sys.stdin.read()); e=d['emulated_ranks']
print('$1', d['ms_per_step'], e['rank_critical_path_ms'], e['projected_speedup_without_communication'])
print('   ', [ (r['prepass'], r['stream_pass'], r['collect'], r['sum_blocks']) for r in e['per_rank_ms']])
" || tail -5 $O/$1.log; }
run a libmodes_gpu.so
run b libmodes_gpu.so
run c libmodes_gpu.so
