#!/bin/bash
tag=${1:-s1}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
p() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['stage_ms'])"; }
timeout 400 python tools/dev_check.py > $out/dev_check.log 2>&1; echo "dev_check rc=$?"; grep -c "^OK" $out/dev_check.log; grep -v "^OK" $out/dev_check.log | head -5
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_formats.py tests/test_gpu_large.py tests/test_gpu_golden.py tests/test_gpu_kernel_generations.py -x -q --timeout 300 > $out/pytest_sel.log 2>&1; tail -3 $out/pytest_sel.log
for i in 1 2; do
echo "== tile 2048"; MGPU_LIBRARY=libmodes_gpu_t2048.so timeout 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | p
echo "== tile 1024"; timeout 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | p
done
R=$(pwd); cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/stats -o bench -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $R/$out/bench_under_rocprof.log 2>&1
cd $R; python - <<'PY'
import csv
for r in csv.DictReader(open('gpurun_out/s34/stats/bench_kernel_stats.csv')):
    print(r['Name'][:40].ljust(42), r['Calls'].rjust(5), f"{float(r['AverageNs'])/1000:9.1f} us  min {float(r['MinNs'])/1000:7.1f}")
PY
