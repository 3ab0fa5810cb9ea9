#!/bin/bash
tag=${1:-s1}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_gather_c.py tests/test_gpu_shard.py -x -q --timeout 600 > $out/pytest_new.log 2>&1; tail -25 $out/pytest_new.log
echo "== config 5, one GPU (nccl, 1 rank)"
MASTER_ADDR=127.0.0.1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29561 bench.py --gpus 1 --config 5 --steps 3 --warmup 1 2>$out/c5.err | tail -1 | cut -c1-900; tail -3 $out/c5.err | grep -v socket
