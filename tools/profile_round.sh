#!/bin/bash
# rocprofv3 evidence for bench.py's roofline numbers: kernel trace + stats, then one --pmc pass per
# counter (FETCH_SIZE, WRITE_SIZE) as MI355X_MICROARCH.md prescribes, then the SQ counters.  Run on the GPU box via gpurun;
# copy the summaries from gpurun_out/ into profiles/ afterwards.
#   usage: tools/profile_round.sh <tag>
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=${1:-r06}
out=$R/gpurun_out/$tag
mkdir -p $out
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o bench -- python $R/bench.py --steps 5 --warmup 1 --loops 8 --no-cpu-baseline --no-extra-configs --event-bracket-us ${BRACKET_US:-4.0} > $out/bench_under_rocprof.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -o bench -- python $R/bench.py --steps 2 --warmup 1 --loops 2 --no-cpu-baseline --no-extra-configs --event-bracket-us ${BRACKET_US:-4.0} > $out/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -o bench -- python $R/bench.py --steps 2 --warmup 1 --loops 2 --no-cpu-baseline --no-extra-configs --event-bracket-us ${BRACKET_US:-4.0} > $out/pmc_write.log 2>&1
timeout 300 rocprofv3 --kernel-trace -i $R/tools/pmc_sq.txt --output-format csv -d $out/pmc_sq -o bench -- python $R/bench.py --steps 1 --warmup 0 --loops 1 --no-cpu-baseline --no-extra-configs --event-bracket-us ${BRACKET_US:-4.0} > $out/pmc_sq.log 2>&1
for i in 0 1; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_extra$i -o extra -- python $R/tools/profile_extra.py $i > $out/extra$i.log 2>&1
done
cd $R
# the summaries carry the hash of the device code they were collected with: bench.py reports their numbers only for that code
sha=$(python -c "import bench; print(bench.kernel_source_sha())")
{ echo "# kernel_source_sha: $sha"; python tools/pmc_kernels.py $out/pmc_sq; } > $out/pmc_sq_summary.txt 2>&1
python tools/pmc_summary.py $(find $out/pmc_fetch -name "*counter_collection.csv" | head -1) $(find $out/pmc_write -name "*counter_collection.csv" | head -1) $out/pmc_hbm.json $sha > /dev/null 2> $out/pmc_hbm.err
timeout 400 python bench.py --steps 20 --warmup 3 > $out/bench_plain.log 2>&1
# the micro harnesses (make -C tools micro): k_sweep alone over a cold > 1 GiB array (and its stages), instruction issue rates, atomics; k_slice's parts
for v in "" _s1 _s2 _s3 _d0 _b1 _b4; do timeout 120 tools/micro/sweep_cold$v 1024 5 4 > $out/sweep_cold$v.json 2> $out/sweep_cold$v.err; done   # _d0: generation 6 (contiguous ranges, no dealer); _b1 / _b4: dealt blocks of 1 / 4 steps
timeout 120 tools/micro/sweep_cold 1024 5 4 0 2000 0 0 0 > $out/sweep_cold_unpaced.json 2>/dev/null
timeout 120 tools/micro/valu_issue $out/valu_issue.json > $out/valu_issue.txt 2>&1
timeout 120 tools/micro/atomic_cost > $out/atomic_cost.txt 2>&1
timeout 300 bash tools/slice_stages.sh > $out/slice_stages.txt 2>&1
timeout 400 bash tools/slice_stage_pmc.sh $tag/stage_pmc > /dev/null 2>&1; cp $out/stage_pmc/stage_pmc.txt $out/slice_stage_pmc.txt 2>/dev/null   # k_slice's instruction counters per left-out stage
tail -1 $out/bench_plain.log | cut -c1-3000
cat $out/stats/bench_kernel_stats.csv | cut -c1-120
cat $out/pmc_hbm.json | head -40
for i in 0 1; do tail -1 $out/extra$i.log | cut -c1-600; cut -c1-110 $out/stats_extra$i/extra_kernel_stats.csv | head -8; done
for v in "" _s1 _s2 _s3 _d0 _b1 _b4 _unpaced; do echo "sweep_cold$v: $(cut -c1-400 $out/sweep_cold$v.json)"; done
tail -12 $out/valu_issue.txt; tail -8 $out/atomic_cost.txt; cat $out/slice_stages.txt; cut -c1-200 $out/slice_stage_pmc.txt
# the whole GPU suite on the code that was profiled
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -5 > $out/gpu_suite_tail.txt; cat $out/gpu_suite_tail.txt
