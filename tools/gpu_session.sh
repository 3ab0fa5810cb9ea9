#!/bin/bash
# scratch: one GPU session
cd $GRAFT_REPO_ROOT
tag=${1:-sess}
mkdir -p gpurun_out/$tag
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -5 > gpurun_out/$tag/tests.txt
for i in 1 2; do timeout 300 python bench.py --no-cpu-baseline --no-extra-configs 2>/dev/null | tail -1 > gpurun_out/$tag/bench$i.json; done
KERNEL="void mgpu::k_window_stats" STAGES="0" bash tools/slice_stage_pmc.sh ${tag}_ws
python - <<'PY' $tag
import json,sys
for i in (1,2):
    try:
        d=json.load(open('gpurun_out/%s/bench%d.json'%(sys.argv[1],i)))
        print('bench',i,d['value'],d['stage_ms'],d['kernels']['k_slice']['avg_launch_ms'],d['roofline']['avg_launch_ms'])
    except Exception as e: print('bench',i,'failed',e)
PY
cat gpurun_out/$tag/tests.txt
