#!/bin/bash
cd /root/repo
out=gpurun_out/r05g; mkdir -p $out
MGPU_LIBRARY=libmodes_gpu_dfwin.so timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -x -q 2>&1 | tail -3
summ() { tail -1 $1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())
k=d.get('kernels',{})
print('$1', 'value', d['value'], 'ms/step', d['ms_per_step'], 'sweep us', round(d['roofline']['avg_launch_ms']*1e3,1), {n:round(v['avg_launch_ms']*1e3,1) for n,v in k.items()}, d.get('stage_ms'))"; }
for i in 1 2; do
timeout 300 python bench.py --steps 20 --warmup 5 --no-extra-configs --no-cpu-baseline > $out/bench_base$i.log 2>/dev/null; summ $out/bench_base$i.log
MGPU_LIBRARY=libmodes_gpu_dfwin.so timeout 300 python bench.py --steps 20 --warmup 5 --no-extra-configs --no-cpu-baseline > $out/bench_dfwin$i.log 2>/dev/null; summ $out/bench_dfwin$i.log
done
