#!/bin/bash
cd /root/repo
out=gpurun_out/r05m; mkdir -p $out
export TMPDIR=/tmp
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$out/stats -o gate -- python /root/repo/tools/bench_gate.py > /root/repo/$out/bench_gate.txt 2> /root/repo/$out/bench_gate.err )
cat $out/bench_gate.txt; tail -3 $out/bench_gate.err
python3 - <<'PY'
import csv
for r in csv.DictReader(open('/root/repo/gpurun_out/r05m/stats/gate_kernel_stats.csv')):
    if 'gate' in r['Name']: print(r['Name'][:50].ljust(50), r['Calls'].rjust(4), ('%.1f'%(float(r['AverageNs'])/1e3)).rjust(9), ('%.1f'%(float(r['MaxNs'])/1e3)).rjust(9))
PY
