#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_wk -o wk -- python /root/repo/tools/wk_probe.py > /root/repo/gpurun_out/s50.out 2>&1
grep "differences" /root/repo/gpurun_out/s50.out | tail -2 | cut -c1-200
t=$(find /tmp/prof_wk -name "*kernel_trace.csv" | head -1)
if [ -n "$t" ]; then python3 - "$t" > /root/repo/gpurun_out/s50_walk_trace.txt <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
t0=int(rows[0]['Start_Timestamp'])
for r in rows:
    if 'walk' in r['Kernel_Name'] or 'slice' in r['Kernel_Name']:
        print(f"{(int(r['Start_Timestamp'])-t0)/1000:9.1f} {(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1000:8.1f} us  {r['Kernel_Name'][:40]}")
PY
tail -26 /root/repo/gpurun_out/s50_walk_trace.txt
fi
