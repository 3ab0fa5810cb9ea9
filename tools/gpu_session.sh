#!/bin/bash
# What one gpurun call of this round usually ran.  (Scratch: edited per call.)
cd /root/repo
mkdir -p gpurun_out/r04b
echo skip tests
timeout 600 python bench.py --config 5 --emulate-ranks 8 --samples 8640000000 --steps 3 --warmup 1 > gpurun_out/r04b/config5_emulate8.json 2> gpurun_out/r04b/config5_emulate8.err
tail -c 3000 gpurun_out/r04b/config5_emulate8.err
python - <<'PY'
import json
try:
    o = json.loads(open("gpurun_out/r04b/config5_emulate8.json").read().strip().splitlines()[-1])
    print("unsharded", o["value"], o["ms_per_step"], "msgs", o["messages_per_step"])
    e = o["emulated_ranks"]
    for k in ("rank_critical_path_ms", "protocol", "rank0_serial_ms", "rank0_serial_total_ms", "rank0_serial_share_of_unsharded", "projected_ms_without_communication", "projected_speedup_without_communication", "gather_bytes_to_rank0", "allgather_bytes_per_round"):
        print(k, e[k])
    for r, p in enumerate(e["per_rank_ms"]):
        print(r, p)
    print(o.get("cpu_baseline"))
except Exception as ex:
    print("no line:", ex)
PY
