#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/s61
SECONDS=0; timeout 900 python bench.py > gpurun_out/s61/bench.log 2> gpurun_out/s61/bench.err; echo "Elapsed $SECONDS s"
tail -1 gpurun_out/s61/bench.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline']['valu_issue'])
for k,v in d.get('configs',{}).items(): print(k, v.get('msamples_s'), v.get('ms_per_segment'), v.get('us_per_launch'), v.get('stage_ms'))"
