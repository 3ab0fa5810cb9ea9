#!/bin/bash
# k_slice's instruction counts with one part left out at a time (experiments build, MGPU_DEBUG_STAGE as in slice_stages.sh;
# 6 = no DF stage, hence no slicing and no scoring either): one counter pass per stage -> gpurun_out/<tag>/stage_pmc.txt
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/${1:-stagepmc}
mkdir -p $out
cd /tmp
for st in ${STAGES:-0 1 2 5 6}; do
  MGPU_LIBRARY=libmodes_gpu_exp.so MGPU_DEBUG_STAGE=$st timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY \
    --output-format csv -d $out/st$st -o bench -- python $R/bench.py --steps 1 --warmup 0 --loops 1 --no-cpu-baseline --no-extra-configs > $out/st$st.log 2>&1
  python - $out/st$st $st ${KERNEL:-mgpu::k_slice} <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if r['Kernel_Name'].startswith(sys.argv[3] if len(sys.argv) > 3 else 'mgpu::k_slice'):
            a = acc[r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
print('stage', sys.argv[2], ' '.join('%s=%.0f' % (k, v[0] / max(v[1], 1)) for k, v in sorted(acc.items())), 'n=%d' % max([v[1] for v in acc.values()] + [0]))
PY
done | tee $out/stage_pmc.txt
