#!/usr/bin/env python3
"""One of bench.py's extra configurations on its own (for `rocprofv3 --kernel-trace --stats -- python tools/profile_extra.py <index>`)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))) if "__file__" in globals() else os.getcwd()
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
import helpers  # noqa: E402

helpers.ensure_built()
name = list(bench.EXTRA_CONFIGS)[int(sys.argv[1])]
fmt, nfix, kw = bench.EXTRA_CONFIGS[name]
print(json.dumps({name: bench.run_extra_config(name, fmt, nfix, kw, 4096 * bench.BUF, 0, bracket_us=4.0, chunk_buffers=2048, ahead=2)}))
