mkdir -p gpurun_out/r06x; O=$GRAFT_REPO_ROOT/gpurun_out/r06x; R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -i $R/tools/pmc_sq.txt --output-format csv -d $O/pmc_sq -o bench -- python $R/bench.py --steps 1 --warmup 0 --loops 1 --no-cpu-baseline --no-extra-configs --event-bracket-us 4.0 > $O/pmc_sq.log 2>&1
cd $R; python tools/pmc_kernels.py $O/pmc_sq > $O/pmc_sq_summary.txt 2>&1; rm -rf $O/pmc_sq
grep -A30 '^k_sweep_uc8' $O/pmc_sq_summary.txt | head -40
