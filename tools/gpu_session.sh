#!/bin/bash
# scratch: the command of one GPU session (gpurun -- 'bash tools/gpu_session.sh'); edit, run, read gpurun_out/
cd /root/repo
out=gpurun_out/r05b; mkdir -p $out
for v in "" _d0 _s1; do timeout 120 tools/micro/sweep_cold$v 1024 5 4 > $out/sweep_cold$v.json 2> $out/sweep_cold$v.err; echo "sweep_cold$v rc=$?: $(cut -c1-1100 $out/sweep_cold$v.json)"; done
timeout 300 bash tools/slice_stages.sh > $out/slice_stages.txt 2>&1; cat $out/slice_stages.txt
export TMPDIR=/tmp
( cd /tmp && timeout 300 rocprofv3 --kernel-trace -i /root/repo/tools/pmc_sq.txt --output-format csv -d /root/repo/$out/pmc_sq -o bench -- python /root/repo/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra-configs --event-bracket-us 3.7 > /root/repo/$out/pmc_sq.log 2>&1 )
python tools/pmc_kernels.py $out/pmc_sq > $out/pmc_sq_summary.txt 2>&1; grep -A22 "^k_slice\|^k_sweep" $out/pmc_sq_summary.txt | head -60
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
