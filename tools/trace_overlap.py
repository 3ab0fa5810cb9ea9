#!/usr/bin/env python3
"""Timeline of the pipeline's kernels from a rocprofv3 --kernel-trace CSV: for a window of the run, every kernel's start / end
(us, relative) and which kernels it overlapped.   usage: tools/trace_overlap.py <kernel_trace.csv> [skip_fraction] [count]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.6
count = int(sys.argv[3]) if len(sys.argv) > 3 else 60
ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("mgpu::", ""), r.get("Queue_Id", "?")) for r in rows))
ks = ks[int(len(ks) * skip):][:count]
t0 = ks[0][0]
for i, (a, b, name, q) in enumerate(ks):
    ov = [n2[:14] for (a2, b2, n2, _) in ks if (a2, b2, n2) != (a, b, name) and a2 < b and b2 > a]
    print(f"{(a - t0) / 1e3:9.1f} {(b - t0) / 1e3:9.1f} {(b - a) / 1e3:7.1f}  q{q:>3} {name[:34]:34s} | {' '.join(ov)}")
