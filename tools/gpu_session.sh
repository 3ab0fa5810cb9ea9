#!/bin/bash
# scratch: the command of one GPU session (gpurun -- 'bash tools/gpu_session.sh'); edit, run, read gpurun_out/
cd /root/repo
out=gpurun_out/r05a; mkdir -p $out
timeout 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_formats.py -x -q 2>&1 | tail -15
for v in "" _d0 _t1 _d0t1; do timeout 120 tools/micro/sweep_cold$v 1024 5 4 > $out/sweep_cold$v.json 2> $out/sweep_cold$v.err; echo "sweep_cold$v rc=$?: $(cut -c1-900 $out/sweep_cold$v.json)"; done
summ() { tail -1 $1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())
k=d.get('kernels',{})
print('$1', 'value', d['value'], 'ms/step', d['ms_per_step'], 'sweep raw us', round(d['roofline']['avg_launch_ms_between_events']*1e3,1), 'frac', d['roofline']['frac'], {n:round(v['avg_launch_ms']*1e3,1) for n,v in k.items()}, d.get('stage_ms'))"; }
timeout 300 python bench.py --steps 20 --warmup 5 --no-extra-configs > $out/bench_new.log 2> $out/bench_new.err; summ $out/bench_new.log
MGPU_LIBRARY=libmodes_gpu_r4.so timeout 300 python bench.py --steps 20 --warmup 5 --no-extra-configs --no-cpu-baseline > $out/bench_r4.log 2> $out/bench_r4.err; summ $out/bench_r4.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-extra-configs --no-cpu-baseline > $out/bench_new2.log 2> $out/bench_new2.err; summ $out/bench_new2.log
tail -3 $out/bench_new.err
