#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/s57
O=gpurun_out/s57
M=tools/micro
for args in "512 9 4 0 2000 0" "512 9 4 1 8000 0" "512 9 4 0 2000 0 0 0" "5 2 2 1 2000 0 1025" "3 2 2 0 2000 0 7169"; do
  echo "== sweep_cold $args" | tee -a $O/sweep_cold.log
  timeout 120 $M/sweep_cold $args 2>&1 | tail -1 | cut -c200-1100 | tee -a $O/sweep_cold.log
done
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -4 | tee $O/pytest.log
timeout 600 python bench.py --steps 10 --warmup 2 --no-extra-configs 2>$O/bench.err | tail -1 > $O/bench.json; cut -c1-1500 $O/bench.json
