#!/bin/bash
# usage: pmc_run.sh <tag> [env assignments...]   -> gpurun_out/pmc_<tag>/
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1; shift
cd /tmp
env "$@" rocprofv3 --kernel-trace -i $R/tools/${PMCFILE:-pmc_sq.txt} --output-format csv -d $R/gpurun_out/pmc_$tag -o bench -- python $R/bench.py --steps 1 --warmup 0 --loops 1 --no-cpu-baseline > $R/gpurun_out/pmc_$tag.log 2>&1
python - "$R/gpurun_out/pmc_$tag" <<'PY'
import csv, glob, sys
for f in sorted(glob.glob(sys.argv[1] + '/pmc_*/bench_counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        if r['Kernel_Name'].startswith(('k_sweep', 'k_slice')):
            print(sys.argv[1].split('/')[-1], r['Counter_Name'], float(r['Counter_Value']))
PY
