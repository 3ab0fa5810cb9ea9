#!/bin/bash
# one GPU-box session: whatever the current investigation needs, outputs under gpurun_out/<tag>/
tag=${1:-s1}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
R=$(pwd)
echo "== dev_check"
timeout 400 python tools/dev_check.py > $out/dev_check.log 2>&1; echo "rc=$?"; grep -c "^OK" $out/dev_check.log; grep -v "^OK" $out/dev_check.log | head
echo "== bench"
timeout 400 python bench.py --steps 10 --warmup 3 > $out/bench.log 2>&1; tail -1 $out/bench.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['stage_ms'], d.get('cpu_baseline',{}).get('bit_identical_to_gpu'))"
for wt in 5 6; do
echo "== bench MGPU_WALK_THREADS=$wt"
MGPU_WALK_THREADS=$wt timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $out/bench_wt$wt.log 2>&1; tail -1 $out/bench_wt$wt.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['stage_ms'])"
done
echo "== kernel stats"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/stats -o bench -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $R/$out/bench_under_rocprof.log 2>&1
cd $R
cat $out/stats/bench_kernel_stats.csv | cut -c1-150 | head -20
