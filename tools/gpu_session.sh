#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/s65
O=gpurun_out/s65

export MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1
timeout 300 python bench.py --config 5 --samples $((4096*131072)) --steps 2 --warmup 1 > $O/c5.out 2>$O/c5.err; echo "rc=$?"
tail -1 $O/c5.out | cut -c1-1800
grep -v "Warning\|amdgpu.ids\|hostname" $O/c5.err | tail -8 | cut -c1-300
