#!/usr/bin/env python3
"""8 fan-in streams as one process of eight against two processes of four (and four of two), run side by side: is the drop per process or per device?"""
import os
import subprocess
import sys
import tempfile
import time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers  # noqa: E402

helpers.ensure_built()
cli = os.path.join(ROOT, "readsb_amd", "host", "readsb_gpu_fanin")
with tempfile.TemporaryDirectory(dir="/dev/shm") as d:
    paths = []
    for k in range(8):
        iq = helpers.synth(seconds=300.0, seed=500 + k, rate=2000.0, threads=32)
        p = os.path.join(d, f"s{k}.iq")
        iq.tofile(p)
        paths.append(p)
        del iq

    def run(groups, tag):
        procs = []
        t0 = time.time()
        for gi, g in enumerate(groups):
            args = [cli]
            for p in g:
                args += ["--ifile", p]
            args += ["--out-prefix", os.path.join(d, f"out{gi}"), "--stats", "--gpu-chunk-buffers", "512"]
            procs.append(subprocess.Popen(args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
        outs = [p.communicate(timeout=600) for p in procs]
        wall = time.time() - t0
        lines = [[ln for ln in e.splitlines() if ln.startswith("fan-in:")][-1] for _, e in outs]
        print(f"== {tag}: wall {wall:.2f} s (process start to end, all processes)")
        for ln in lines:
            print("   ", ln)

    for rep in range(2):
        run([paths], "1 process x 8 streams")
        run([paths[:4], paths[4:]], "2 processes x 4 streams")
        run([paths[0:2], paths[2:4], paths[4:6], paths[6:8]], "4 processes x 2 streams")
        run([paths[:4]], "1 process x 4 streams")
