#!/bin/bash
# timing experiment: k_sweep_slice with later stages disabled (results are NOT valid messages)
for st in 0 1 16; do
  MGPU_DEBUG_STAGE=$st python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 > /tmp/line.json
  python - "$st" <<'PY'
import json, sys
d = json.load(open('/tmp/line.json'))
print("debug_stage", sys.argv[1], "sweep_slice_ms", d["stage_ms"]["sweep_slice"], "value", d["value"])
PY
done
