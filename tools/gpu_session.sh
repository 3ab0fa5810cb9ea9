#!/bin/bash
# scratch: the command of one GPU session (gpurun -- 'bash tools/gpu_session.sh'); edit, run, read gpurun_out/
cd /root/repo
out=gpurun_out/r05s; mkdir -p $out
uptime > $out/uptime.txt
timeout 600 python bench.py > $out/bench_default.log 2> $out/bench_default.err
tail -1 $out/bench_default.log | cut -c1-2500
cat $out/uptime.txt
