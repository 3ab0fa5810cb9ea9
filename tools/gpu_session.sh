#!/bin/bash
cd /root/repo
out=gpurun_out/r05f; mkdir -p $out
timeout 300 python -m pytest tests/test_gpu_gate.py -x -q 2>&1 | tail -3
summ() { tail -1 $1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())
k=d.get('kernels',{})
print('$1', 'value', d['value'], 'ms/step', d['ms_per_step'], 'sweep us', round(d['roofline']['avg_launch_ms']*1e3,1), {n:round(v['avg_launch_ms']*1e3,1) for n,v in k.items()}, d.get('stage_ms'))"; }
timeout 300 python bench.py --steps 20 --warmup 5 --no-extra-configs --no-cpu-baseline > $out/bench_plain.log 2> $out/bench_plain.err; summ $out/bench_plain.log
timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --exercise-gather --steps 20 --warmup 5 --no-extra-configs --no-cpu-baseline > $out/bench_gather.log 2> $out/bench_gather.err; summ $out/bench_gather.log; tail -3 $out/bench_gather.err
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_extras.log 2> $out/bench_extras.err; tail -1 $out/bench_extras.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('headline', d['value'])
for k,v in d['configs'].items():
    print(k, v.get('msamples_s'), v.get('msamples_s_both_repetitions'), v.get('us_per_launch'), v.get('stage_ms'))"
