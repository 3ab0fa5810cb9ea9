#!/bin/bash
cd /root/repo
O=gpurun_out/r04n
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_shard.py -m gpu -x -q -k "stream" 2>&1 | tail -8
timeout 600 python bench.py --config 5 --emulate-ranks 8 --samples 8640000000 --steps 3 --warmup 1 --no-cpu-baseline > $O/config5_stream8.json 2> $O/config5_stream8.err
tail -c 1500 $O/config5_stream8.err
python - <<'PY'
import json
try:
    o = json.loads(open("gpurun_out/r04n/config5_stream8.json").read().strip().splitlines()[-1])
    print("unsharded", o["value"], o["ms_per_step"], "msgs", o["messages_per_step"])
    e = o["emulated_ranks"]
    for k in ("form", "rank_critical_path_ms", "protocol", "rank0_serial_ms", "rank0_serial_share_of_unsharded", "projected_ms_without_communication", "projected_speedup_without_communication"):
        print(k, e[k])
    for r, p in enumerate(e["per_rank_ms"]):
        print(r, p)
except Exception as ex:
    print("no line:", ex)
PY
