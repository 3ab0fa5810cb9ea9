#!/bin/bash
# scratch: the command of one GPU session (gpurun -- 'bash tools/gpu_session.sh'); edit, run, read gpurun_out/
cd /root/repo
out=gpurun_out/r05d; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_formats.py tests/test_gpu_deferred.py tests/test_gpu_modeac.py -x -q 2>&1 | tail -5
{ lscpu | grep -i "model name\|socket\|numa\|thread\|^CPU(s)"; nproc; cat /sys/class/drm/card*/device/numa_node 2>/dev/null | head -3; grep -i "amdgpu\|kfd" /proc/interrupts | awk '{s=0; for(i=2;i<=NF-3;i++) if ($i+0>0) {printf "cpu%d:%s ", i-2, $i}; print $NF}' | head -20; uptime; } > $out/topology.txt 2>&1
summ() { tail -1 $1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())
k=d.get('kernels',{})
print('$1', 'value', d['value'], 'ms/step', d['ms_per_step'], 'sweep raw us', round(d['roofline']['avg_launch_ms_between_events']*1e3,1), {n:round(v['avg_launch_ms']*1e3,1) for n,v in k.items()}, d.get('stage_ms'))"; }
for i in 1 2 3; do timeout 300 python bench.py --steps 20 --warmup 5 --no-extra-configs --no-cpu-baseline > $out/bench_$i.log 2> $out/bench_$i.err; summ $out/bench_$i.log; done
for wt in 6 4; do for bt in 6 3; do MGPU_LIBRARY=libmodes_gpu_exp.so MGPU_WALK_THREADS=$wt MGPU_BUILD_THREADS=$bt timeout 300 python bench.py --steps 20 --warmup 5 --no-extra-configs --no-cpu-baseline > $out/bench_w${wt}b${bt}.log 2>/dev/null; summ $out/bench_w${wt}b${bt}.log; done; done
grep -i "amdgpu\|kfd" /proc/interrupts | awk '{for(i=2;i<=NF-3;i++) if ($i+0>0) {printf "cpu%d:%s ", i-2, $i}; print $NF}' | head -20 > $out/irq_after.txt
cat $out/topology.txt | head -30; echo; cat $out/irq_after.txt | cut -c1-600
