#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/dump
MGPU_DUMP_DIR=gpurun_out/dump timeout 300 python bench.py --no-extra-configs --no-cpu-baseline --steps 1 --warmup 0 2>&1 | tail -1 | cut -c1-200
ls -la gpurun_out/dump
