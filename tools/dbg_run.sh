#!/bin/bash
# timing experiments with stages of k_sweep_slice disabled (results are not valid messages).
# every run is wrapped in `timeout` so a bad debug combination cannot hang the box.
for st in "$@"; do
  echo "== MGPU_DEBUG_STAGE=$st"
  MGPU_DEBUG_STAGE=$st MGPU_DEBUG_PRINT=1 timeout 90 python bench.py --steps 1 --warmup 1 --no-cpu-baseline 2>&1 | grep -E "dbg:|Error|error" | tail -3
done
