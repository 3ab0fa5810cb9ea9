#!/bin/bash
# A/B of k_sweep_uc8 variants alone: tools/ab/ab_uc8v.sh <tag> <reps> <suffix>...
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/$1; shift
reps=$1; shift
mkdir -p $out
cd /tmp
timeout 100 $R/tools/micro/sweep_uc8_cold 2048 3 300 > /dev/null 2>&1    # (clocks up)
for rep in $(seq $reps); do for v in "" "$@"; do
  timeout 300 $R/tools/micro/sweep_uc8_cold$v 2048 3 150 > $out/v${v}_$rep.json 2>> $out/err.txt
  echo "variant '$v' rep $rep: $(python3 -c "import json; d=json.load(open('$out/v${v}_$rep.json')); u=d['us']; print(u['min'], u['median'], u['mean'], d['frac_of_8TBs'], d['magnitude_mismatches_vs_cpu_table']+d['buffer_sum_mismatches']+d['candidate_mismatches_vs_cpu_scan'])")"
done; done
tail -3 $out/err.txt
