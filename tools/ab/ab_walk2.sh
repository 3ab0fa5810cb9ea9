mkdir -p gpurun_out/$1; O=gpurun_out/$1
MGPU_LIBRARY=libmodes_gpu_exp.so MGPU_DEBUG_PRINT=1 timeout 300 python bench.py --no-cpu-baseline --no-extra-configs --steps 1 --warmup 1 --loops 6 2> $O/dbg.txt > /dev/null; grep -E "dbg: (walk|[0-9]+ round)" $O/dbg.txt | tail -8
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_deferred.py tests/test_gpu_shard.py -x -q 2>&1 | tail -3
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --no-cpu-baseline --no-extra-configs --steps 20 > $O/$tag.log 2>&1; tail -1 $O/$tag.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['stage_ms']; print('$tag', d['value'], d.get('ms_per_feed'), s['sweep'], s['slice'], s['prescreen'], 'host', s['d2h'], s['resolve_host'], s['build_host'], s['build_wait'], d['roofline']['frac'])" 2>/dev/null || tail -5 $O/$tag.log; }
for i in 1 2 3 4; do
run main_$i X=1
done
# does rocprofv3 keep the side streams' CU masks?  k_sweep_uc8's average with the mask (default) and without
cd /tmp; export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-/root/repo}
for m in 8 0; do
  MGPU_LIBRARY=libmodes_gpu_exp.so MGPU_CU_MASK_STRIDE=$m timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/rp_mask$m -o b -- python $R/bench.py --steps 3 --warmup 1 --loops 8 --no-cpu-baseline --no-extra-configs --event-bracket-us 3.7 > $R/$O/rp_mask$m.log 2>&1
  echo "mask stride $m:"; grep -E "k_sweep_uc8|k_slice|copyBuffer" $R/$O/rp_mask$m/b_kernel_stats.csv | cut -c1-120
done
