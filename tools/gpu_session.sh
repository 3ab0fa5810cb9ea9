#!/bin/bash
cd /root/repo
out=gpurun_out/r05k; mkdir -p $out
timeout 900 python bench.py --config 5 --emulate-ranks 8 --samples 8640000000 --steps 3 --warmup 1 > $out/c5_hour_cpu.json 2> $out/c5_hour_cpu.err
tail -1 $out/c5_hour_cpu.json | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())
e=d['emulated_ranks']
print('value', d['value'], 'ms/step', d['ms_per_step'])
print('per rank', e['per_rank_ms'])
print('critical', e['rank_critical_path_ms'], 'projected', e['projected_ms_without_communication'], e['projected_speedup_without_communication'], 'identical', e['identical_to_unsharded'], 'rank0 serial', e['rank0_serial_total_ms'])
print(d.get('cpu_baseline'))"
tail -3 $out/c5_hour_cpu.err
