#!/bin/bash
cd /root/repo
R=/root/repo
mkdir -p gpurun_out/r02f
timeout 300 python -m pytest tests/test_gpu_modeac.py tests/test_gpu_pipeline_chain.py tests/test_gpu_device_walk.py tests/test_gpu_large.py -x -q 2>&1 | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 | cut -c1-120
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02f/stats2 -o bench -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extra-configs > $R/gpurun_out/r02f/bench_under_rocprof.log 2>&1
cd $R
f=$(find gpurun_out/r02f/stats2 -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cp "$f" gpurun_out/r02f/kernel_stats.csv; cut -c1-100 "$f" | head -14; fi
timeout 600 python bench.py > gpurun_out/r02f/bench_plain.log 2> gpurun_out/r02f/bench_plain.err
tail -1 gpurun_out/r02f/bench_plain.log > gpurun_out/r02f/bench_line.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r02f/bench_line.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], d.get('stage_ms'))
print(d.get('cpu_baseline',{}).get('value'), d.get('pcie_inclusive_msamples_s'))
for k,v in d.get('configs',{}).items(): print(k, v['msamples_s'], v['ms_per_segment'], v.get('device_walk'))
PY
