# do several ranks' host pipelines coexist on one socket?  4 processes x 1 context, pinned as LOCAL_RANK 0-3 would be, all on GPU 0
# (the rates mean nothing: the GPU is shared; the host stages' own wall time is the point) — profiles/r06_host_4rank.txt
mkdir -p gpurun_out/$1; O=gpurun_out/$1
A="--steps 5 --warmup 2 --loops 8 --samples 134217728 --no-cpu-baseline --no-extra-configs"
timeout 300 python bench.py $A > $O/one.log 2>&1
tail -1 $O/one.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('1 rank ', d['value'], d['ms_per_feed'], d['stage_ms'])"
timeout 600 python bench.py --gpus 4 --dryrun-gloo $A > $O/four.log 2>&1
tail -1 $O/four.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('4 ranks', d['value'], d['ms_per_feed'], d['stage_ms']); [print('   ', r) for r in d['per_rank_host_ms']]" || tail -5 $O/four.log
