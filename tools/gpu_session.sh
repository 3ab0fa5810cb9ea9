#!/bin/bash
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_gate.py -x -q -s 2>&1 | tail -25
