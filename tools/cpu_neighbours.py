#!/usr/bin/env python3
"""Who else is on this node's CPUs?  Per-CPU busy fraction from /proc/stat over an interval while THIS container is idle, grouped by
L3 group, with the groups the pipeline of device 0 would pin to marked.   usage: tools/cpu_neighbours.py [seconds=1.0]"""
import collections
import sys
import time


def snap():
    out = {}
    for l in open("/proc/stat"):
        if l.startswith("cpu") and l[3].isdigit():
            f = l.split()
            v = list(map(int, f[1:]))
            out[int(f[0][3:])] = (sum(v), v[3] + v[4])     # total, idle + iowait
    return out


def l3_of(cpu):
    try:
        return int(open(f"/sys/devices/system/cpu/cpu{cpu}/cache/index3/id").read())
    except OSError:
        return -1


dt = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
a = snap()
time.sleep(dt)
b = snap()
busy = {c: 1.0 - (b[c][1] - a[c][1]) / max(1, b[c][0] - a[c][0]) for c in a}
groups = collections.defaultdict(list)
for c in sorted(busy):
    groups[l3_of(c)].append(c)
print("loadavg", open("/proc/loadavg").read().strip(), "cpus", len(busy))
for g in sorted(groups):
    cs = groups[g]
    bs = [busy[c] for c in cs]
    print(f"L3 {g:3d}: cpus {cs[0]:3d}..{cs[-1]:3d} ({len(cs)})  mean busy {sum(bs) / len(bs):.2f}  max {max(bs):.2f}  busy>0.5: {sum(x > 0.5 for x in bs)}")
