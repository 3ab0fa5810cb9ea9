#!/bin/bash
# scratch: what the last GPU session of the round ran (gpurun -- 'bash tools/gpu_session.sh')
cd /root/repo
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_kernel_generations.py tests/test_gpu_prescreen_chains.py -m gpu -x -q 2>&1 | tail -3
timeout 600 python bench.py --steps 20 --warmup 3 > gpurun_out/bench_keys.log 2>&1; tail -1 gpurun_out/bench_keys.log | cut -c1-120
