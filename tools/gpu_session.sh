#!/bin/bash
cd /root/repo
O=gpurun_out/r04q
mkdir -p $O
for w in 0 1 0 1; do
  if [ $w = 1 ]; then export MGPU_X_FSUM_AFTER_SLICE=1; else unset MGPU_X_FSUM_AFTER_SLICE; fi
  timeout 300 python bench.py --steps 5 --warmup 2 > $O/bench_$w.json 2> $O/bench_$w.err
  python - $w <<'PY'
import json, sys
try:
    o = json.loads(open(f"gpurun_out/r04q/bench_{sys.argv[1]}.json").read().strip().splitlines()[-1])
    for k, v in list(o.get("configs", {}).items())[:1]:
        print("after_slice", sys.argv[1], k[:22], v.get("msamples_s_both_repetitions"), v.get("host_stage_ms_both_repetitions"), v.get("us_per_launch"))
except Exception as e:
    print("no line:", e)
PY
done
