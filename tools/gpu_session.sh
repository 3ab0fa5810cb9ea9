#!/bin/bash
cd /root/repo
timeout 1200 python -m pytest tests/test_gpu_shard_c.py -m gpu -x -q 2>&1 | tail -15
