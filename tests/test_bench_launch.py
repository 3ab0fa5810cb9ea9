"""`bench.py --gpus N` is the launcher: N ranks on one node without torchrun on the command line, and loud errors when
the flag, the environment and the node disagree (round 5 parsed the flag and ignored it: `python bench.py --gpus 8` would
have measured one GPU and said so).  No GPU needed: --dryrun-gloo --launch-check stops after the ranks have met."""
import json
import os
import subprocess
import sys

import helpers

BENCH = os.path.join(helpers.ROOT, "bench.py")


def run(args, env_extra=None, timeout=240):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH] + args, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)


def last_json(r):
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert lines, r.stdout + r.stderr
    return json.loads(lines[-1])


def test_gpus_flag_launches_that_many_ranks():
    r = run(["--gpus", "2", "--dryrun-gloo", "--launch-check"])
    assert r.returncode == 0, r.stderr[-2000:]
    d = last_json(r)
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["gpus_flag"] == 2 and d["launched_by"] == "torchrun"


def test_one_rank_needs_no_launcher():
    r = run(["--gpus", "1", "--dryrun-gloo", "--launch-check"])
    assert r.returncode == 0, r.stderr[-2000:]
    d = last_json(r)
    assert d["n_gpus"] == 1 and d["launched_by"] == "direct"


def test_flag_and_world_size_must_agree():
    r = run(["--gpus", "2", "--dryrun-gloo", "--launch-check"], {"WORLD_SIZE": "4", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE is 4" in r.stderr
    r = run(["--dryrun-gloo", "--launch-check"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})      # default --gpus 1 under a 2-rank launcher
    assert r.returncode != 0 and "--gpus 1" in r.stderr


def test_more_ranks_than_gpus_is_an_error():
    import torch
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    r = run(["--gpus", str(have + 1 if have else 2), "--launch-check"])
    assert r.returncode != 0 and "one rank per GPU" in r.stderr
