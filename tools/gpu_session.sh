#!/bin/bash
cd /root/repo
export TMPDIR=/tmp
R=/root/repo
out=$R/gpurun_out/s51
mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_convert.py tests/test_gpu_modeac.py tests/test_gpu_formats.py tests/test_gpu_large.py -x -q 2>&1 | tail -4
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_extra0 -o extra -- python $R/tools/profile_extra.py 0 > $out/extra0.log 2>&1
cd $R
python3 - <<'PY'
import csv
for r in list(csv.DictReader(open('gpurun_out/s51/stats_extra0/extra_kernel_stats.csv')))[:5]:
    print(r['Name'].split('(')[0][:50].ljust(52), r['Calls'].rjust(4), round(float(r['AverageNs'])/1e3,1), round(float(r['MinNs'])/1e3,1), round(float(r['MaxNs'])/1e3,1))
PY
timeout 300 python tools/profile_extra.py 0 2>/dev/null | tail -1 | cut -c1-420
