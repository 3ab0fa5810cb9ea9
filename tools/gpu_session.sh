#!/bin/bash
# scratch: the command of one GPU session (gpurun -- 'bash tools/gpu_session.sh'); edit, run, read gpurun_out/
cd /root/repo
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
