#!/bin/bash
# scratch: the command of the current `gpurun -- 'bash tools/gpu_session.sh'` call (edited per call during development).
cd /root/repo
mkdir -p gpurun_out/s50
O=gpurun_out/s50
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -4 | tee $O/pytest.log
M=tools/micro
for args in "512 9 4 0 2000 0" "512 9 4 1 8000 0" "512 9 3 0 2000 8192" "512 9 3 0 2000 1024" "512 9 3 0 2000 2048" "480 9 3 0 2000 0"; do
  echo "== sweep_cold $args" | tee -a $O/sweep_cold.log
  timeout 120 $M/sweep_cold $args 2>&1 | tail -1 | tee -a $O/sweep_cold.log
done
for s in 1 2 3; do
  echo "== sweep_cold_s$s" | tee -a $O/sweep_cold.log
  timeout 120 $M/sweep_cold_s$s 512 9 3 0 2000 0 2>&1 | tail -1 | tee -a $O/sweep_cold.log
done
timeout 600 python bench.py --steps 10 --warmup 2 --no-extra-configs 2>$O/bench.err | tail -1 > $O/bench.json; cut -c1-1800 $O/bench.json
MGPU_CHUNK_BUFFERS=480 timeout 600 python bench.py --steps 10 --warmup 2 --no-extra-configs --no-cpu-baseline --samples $((3840*131072)) 2>$O/bench480.err | tail -1 > $O/bench480.json; cut -c1-900 $O/bench480.json
