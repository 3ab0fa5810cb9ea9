#!/bin/bash
cd /root/repo
O=gpurun_out/r04o
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_shard.py -m gpu -x -q 2>&1 | tail -4
for form in stream packets; do
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --config 5 --gpus 2 --dryrun-gloo --samples 1440000000 --steps 2 --warmup 1 --config5-form $form > $O/config5_2rank_$form.json 2> $O/config5_2rank_$form.err
tail -c 600 $O/config5_2rank_$form.err | grep -v "socket.cpp"
python - $form <<'PY'
import json, sys
try:
    o = json.loads(open(f"gpurun_out/r04o/config5_2rank_{sys.argv[1]}.json").read().strip().splitlines()[-1])
    print(sys.argv[1], o["value"], o["ms_per_step"], o["n_gpus"], o.get("rank0_phase_ms_per_step"), o.get("protocol"), (o.get("cpu_baseline") or {}).get("bit_identical_to_gpu"), (o.get("cpu_baseline") or {}).get("counters_compared"))
except Exception as ex:
    print("no line:", ex)
PY
done
