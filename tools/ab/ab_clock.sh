#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/${1:-r06bf}
mkdir -p $out
cd /tmp
T=$R/tools/micro/sweep_uc8_cold
(rocm-smi --showclocks --showpower --showmaxpower 2>&1 | head -40) > $out/smi_idle.txt
bash $R/tools/clock_watch.sh $out/clock_sweep.txt -- timeout 300 $T 2048 3 2000 > $out/sweep_long.json 2> $out/err.txt
bash $R/tools/clock_watch.sh $out/clock_bench.txt -- timeout 300 python $R/bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-extra-configs > $out/bench.log 2>> $out/err.txt
head -3 $out/clock_sweep.txt; awk 'NR%6==0' $out/clock_sweep.txt | head -30; echo; awk 'NR%10==0' $out/clock_bench.txt | tail -30
cut -c1-600 $out/sweep_long.json; tail -1 $out/bench.log | cut -c1-300; cat $out/smi_idle.txt | head -30
