#!/usr/bin/env python3
"""rocprofv3 --kernel-trace CSV -> one short line per kernel (start us, end us, queue, name), sorted by start.
   usage: tools/trace_compact.py <kernel_trace.csv> <out.txt>"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r["Kernel_Name"].split("(")[0].replace("mgpu::", "").replace("void ", "")) for r in rows))
t0 = ks[0][0]
with open(sys.argv[2], "w") as f:
    for a, b, q, name in ks:
        f.write(f"{(a - t0) / 1e3:.1f} {(b - t0) / 1e3:.1f} {q} {name[:40]}\n")
