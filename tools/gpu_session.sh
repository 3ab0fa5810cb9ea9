#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -6 > gpurun_out/s52_suite.log
cat gpurun_out/s52_suite.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
timeout 600 python bench.py 2>gpurun_out/s52_bench.err | tail -1 > gpurun_out/s52_bench.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/s52_bench.json'))
print(d['value'], d['ms_per_step'], d['roofline'], d.get('stage_ms'))
for k,v in d.get('configs',{}).items(): print(k, v['msamples_s'], v['ms_per_segment'], v.get('device_walk'))
PY
