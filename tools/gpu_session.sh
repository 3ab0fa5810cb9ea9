#!/bin/bash
# scratch: the command of the current `gpurun -- 'bash tools/gpu_session.sh'` call (edited per call during development).
# As committed: the round's closing checks — GPU tests, smoke, the default bench line.
cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -4
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 | cut -c1-160
timeout 600 python bench.py 2>gpurun_out/bench.err | tail -1 | cut -c1-400
