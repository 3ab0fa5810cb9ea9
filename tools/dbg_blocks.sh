#!/bin/bash
for b in "$@"; do
  echo "== MGPU_SWEEP_BLOCKS=$b"
  MGPU_SWEEP_BLOCKS=$b MGPU_DEBUG_PRINT=1 timeout 90 python bench.py --steps 1 --warmup 1 --no-cpu-baseline 2>&1 | grep -E "dbg: per|^\{" | tail -2 | cut -c1-330
done
