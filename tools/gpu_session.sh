#!/bin/bash
cd /root/repo
O=gpurun_out/r04m
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_deferred.py tests/test_gpu_device_walk.py tests/test_gpu_shard.py tests/test_gpu_golden.py -m gpu -x -q 2>&1 | tail -3
export MGPU_DBG_BENCH_REPS=4
for i in 0 1; do timeout 300 python tools/profile_extra.py $i 2>/dev/null | tail -1 | python -c "
import json,sys
o=json.loads(sys.stdin.readline())
for k,v in o.items(): print(k[:20], v['msamples_s_both_repetitions'], v['live_records_per_1000_samples'], [ (h['d2h'],h['resolve_host'],h['build_host']) for h in v['host_stage_ms_both_repetitions']][:2])"; done
unset MGPU_DBG_BENCH_REPS
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json, sys
try:
    o = json.loads(open("gpurun_out/r04m/bench.json").read().strip().splitlines()[-1])
    print("headline", round(o["value"]), o["ms_per_step"], o["stage_ms"])
    for k, v in o.get("configs", {}).items():
        print("   ", k[:22], v.get("msamples_s_both_repetitions"), v.get("host_stage_ms_both_repetitions"), v.get("us_per_launch"))
except Exception as e:
    print("no line:", e)
PY
