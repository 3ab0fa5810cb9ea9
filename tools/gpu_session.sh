#!/bin/bash
# What one gpurun call of this round usually ran: the GPU tests, then the benchmark line.  (Scratch: edited per call.)
cd /root/repo
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
timeout 900 python bench.py 2>/dev/null | tail -1 | cut -c1-600
