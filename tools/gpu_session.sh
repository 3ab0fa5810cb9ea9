#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/s43
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
TR="python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511"
timeout 1500 $TR bench.py --gpus 1 --config 5 --samples 8640000000 --steps 3 --warmup 1 > gpurun_out/s43/config5_hour.log 2>&1
tail -1 gpurun_out/s43/config5_hour.log | cut -c1-2500
timeout 300 $TR bench.py --gpus 1 --config 5 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-1500
