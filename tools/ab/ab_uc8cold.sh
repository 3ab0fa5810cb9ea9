#!/bin/bash
# k_sweep_uc8 alone over cold memory (tools/micro/sweep_uc8_cold.hip): by its own HIP events, then the same program under
# rocprofv3 --kernel-trace --stats — the two have to agree.  200 rounds x 3 replicas = 597 timed launches back to back (~0.14 s:
# the clocks have settled), and a short run of isolated launches (a host wait behind each).
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/${1:-r06bg}
mkdir -p $out
cd /tmp
T=$R/tools/micro/sweep_uc8_cold
timeout 300 $T 2048 3 200 > $out/sweep_uc8_cold.json 2> $out/sweep_uc8_cold.err; echo "rc $?"
timeout 300 $T 2048 3 40 0 2000 0 > $out/sweep_uc8_cold_one_at_a_time.json 2>> $out/sweep_uc8_cold.err; echo "rc $?"
timeout 300 $T 2048 3 200 1 > $out/sweep_uc8_cold_dense.json 2>> $out/sweep_uc8_cold.err; echo "rc $?"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o uc8cold -- $T 2048 3 200 > $out/sweep_uc8_cold_under_rocprof.json 2> $out/rocprof.err; echo "rc $?"
for f in $out/sweep_uc8_cold*.json; do echo "$(basename $f): $(python3 -c "import json,sys; d=json.load(open('$f')); print(d['queued'], d['us'], d['frac_of_8TBs'], d['magnitude_mismatches_vs_cpu_table'], d['buffer_sum_mismatches'], d['candidate_mismatches_vs_cpu_scan'])")"; done
cut -c1-160 $out/stats/uc8cold_kernel_stats.csv; tail -5 $out/sweep_uc8_cold.err
rm -f $out/stats/uc8cold_kernel_trace.csv
