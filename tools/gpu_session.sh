#!/bin/bash
cd /root/repo
TR="python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533"
for mode in "" "--exercise-gather" "" "--exercise-gather"; do
timeout 600 $TR bench.py --gpus 1 $mode --no-extra-configs --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$mode', d['value'], d['ms_per_step'], d['stage_ms'])"
done
