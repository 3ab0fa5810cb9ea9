#!/bin/bash
cd /root/repo
out=gpurun_out/r05p; mkdir -p $out
timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_deferred.py tests/test_gpu_formats.py -x -q 2>&1 | tail -2
for i in $(seq 1 16); do
timeout 300 python bench.py --steps 20 --warmup 5 --no-extra-configs --no-cpu-baseline > $out/run$i.log 2> $out/run$i.err
tail -1 $out/run$i.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())
s=d['stage_ms']
print('run $i', 'value', d['value'], 'ms/step', d['ms_per_step'], 'd2h', s['d2h'], 'walk', s['resolve_host'], 'build', s['build_host'], 'sigp', s['sigpower'])"
done
