#!/bin/bash
cd /root/repo
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-/root/repo}
bash tools/profile_round.sh r03d 2>&1 | tail -40 | cut -c1-600
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
