#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05k
python - <<'PY' 2>&1 | tee gpurun_out/r05k/topology.txt
import glob, os, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
rows=[]
for d in sorted(glob.glob('/sys/bus/pci/devices/*')):
    try:
        v=open(d+'/vendor').read().strip(); dev=open(d+'/device').read().strip(); cl=open(d+'/class').read().strip()
        if v=='0x1002' and (cl.startswith('0x12') or cl.startswith('0x03')):
            rows.append((os.path.basename(d), dev, cl, open(d+'/local_cpulist').read().strip(), open(d+'/numa_node').read().strip()))
    except Exception as e: pass
for r in rows: print(r)
import helpers, readsb_amd
d = readsb_amd.Demodulator(max_samples=8*131072)
print('host_cpus', d.host_cpus())
d.close()
PY
for rep in 1 2; do
  timeout 300 python bench.py --no-cpu-baseline --no-extra-configs 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print(d['value'], d['stage_ms'])"
done
timeout 600 python -m pytest tests/test_gpu_fanin.py tests/test_gpu_gather_c.py tests/test_gpu_host_cli.py -x -q -m gpu 2>&1 | tail -2
