#!/bin/bash
cd /root/repo
timeout 900 python -c "
import sys; sys.argv=['bench.py']
import bench
bench.restore_affinity = lambda: None       # the library alone has to get it right
bench.main()" 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())
print(d['value'], d['ms_per_step'])
for k,v in d.get('configs',{}).items(): print(k, v.get('msamples_s'), v.get('ms_per_segment'), v.get('stage_ms'))"
timeout 600 python -m pytest tests/test_gpu_fanin.py tests/test_gpu_deferred.py -x -q 2>&1 | tail -3
