#!/bin/bash
# What one gpurun call of this round usually ran.  (Scratch: edited per call.)
cd /root/repo
O=gpurun_out/r04c
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_full.json 2> $O/bench_full.err
python - <<'PY'
import json
try:
    o = json.loads(open("gpurun_out/r04c/bench_full.json").read().strip().splitlines()[-1])
    r = o["roofline"]
    print(round(o["value"]), o["ms_per_step"], "sweep", r["avg_launch_ms"], r["frac"], "raw", r["avg_launch_ms_between_events"], "slice", o["kernels"]["k_slice"]["avg_launch_ms"], o["stage_ms"], "pcie", o.get("pcie_inclusive_msamples_s"))
    print(o["cpu_baseline"])
    for k, v in o.get("configs", {}).items():
        print(k, {kk: v.get(kk) for kk in ("msamples_s", "ms_per_segment", "us_per_launch", "stage_ms", "error")})
except Exception as e:
    print("no line:", e)
PY
tail -3 $O/bench_full.err
# four ranks' host pipelines on ONE socket (what an 8-GPU node's socket carries), all on GPU 0: do their polling threads coexist?
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 4 --dryrun-gloo --steps 10 --warmup 3 --samples 134217728 --no-cpu-baseline > $O/host_4rank.json 2> $O/host_4rank.err
timeout 300 python bench.py --steps 10 --warmup 3 --samples 134217728 --no-cpu-baseline --no-extra-configs > $O/host_1rank.json 2> $O/host_1rank.err
python - <<'PY'
import json
for f in ("host_4rank", "host_1rank"):
    try:
        o = json.loads(open(f"gpurun_out/r04c/{f}.json").read().strip().splitlines()[-1])
        print(f, round(o["value"]), o["ms_per_step"], o["stage_ms"], o.get("per_rank_host_ms"))
    except Exception as e:
        print(f, "no line:", e)
PY
tail -5 $O/host_4rank.err
