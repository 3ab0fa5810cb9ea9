#!/bin/bash
cd /root/repo
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-/root/repo}
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
bash tools/profile_round.sh r03e 2>&1 | tail -30 | cut -c1-300
