#!/bin/bash
# scratch: what the last GPU session of the round ran (gpurun -- 'bash tools/gpu_session.sh')
export TMPDIR=/tmp
R=/root/repo
out=$R/gpurun_out/r04h
mkdir -p $out
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o bench -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extra-configs --event-bracket-us 4.0 > $out/bench_under_rocprof.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-configs --event-bracket-us 4.0 > $out/pmc_fetch.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-configs --event-bracket-us 4.0 > $out/pmc_write.log 2>&1
timeout 200 rocprofv3 --kernel-trace -i $R/tools/pmc_sq.txt --output-format csv -d $out/pmc_sq -o bench -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra-configs --event-bracket-us 4.0 > $out/pmc_sq.log 2>&1
for i in 0 1; do
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_extra$i -o extra -- python $R/tools/profile_extra.py $i > $out/extra$i.log 2>&1
done
cd $R
sha=$(python -c "import bench; print(bench.kernel_source_sha())")
{ echo "# kernel_source_sha: $sha"; python tools/pmc_kernels.py $out/pmc_sq; } > $out/pmc_sq_summary.txt 2>&1
python tools/pmc_summary.py $(find $out/pmc_fetch -name "*counter_collection.csv" | head -1) $(find $out/pmc_write -name "*counter_collection.csv" | head -1) $out/pmc_hbm.json $sha > /dev/null 2> $out/pmc_hbm.err
cut -c1-100 $out/stats/bench_kernel_stats.csv | head -8
