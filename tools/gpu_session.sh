#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_gather_c.py tests/test_gpu_deferred.py tests/test_gpu_dropin.py tests/test_gpu_shard_c.py -x -q -m gpu 2>&1 | tail -4
timeout 300 python bench.py --no-cpu-baseline --no-extra-configs --steps 1 --warmup 0 2>&1 | tail -1 | cut -c1-300
