#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/$1; shift
mkdir -p $out
cd /tmp
for rep in 1 2 3; do for sh in 0 12 4; do
  UC8_IQ_SHIFT=$sh timeout 300 $R/tools/micro/sweep_uc8_cold 2048 3 200 > $out/s${sh}_$rep.json 2>> $out/err.txt
  echo "shift $sh rep $rep: $(python3 -c "import json; d=json.load(open('$out/s${sh}_$rep.json')); print(d['us'], d['frac_of_8TBs'])")"
done; done
