#!/bin/bash
cd /root/repo
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
python __graft_entry__.py smoke 2>&1 | tail -1 | cut -c1-200
bash tools/profile_round.sh r04 2>&1 | tail -60
