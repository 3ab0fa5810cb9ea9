#!/usr/bin/env python3
"""Static instruction counts per basic block of one kernel in readsb_amd/csrc/kernels.s (make -C readsb_amd/csrc asm).
   usage: tools/isa_blocks.py <mangled-name-substring> [--dump LABEL]"""
import collections
import re
import sys

name = sys.argv[1]
lines = open("readsb_amd/csrc/kernels.s").read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*" + re.escape(name) + r"\S*:", l))
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))   # (a kernel may hold several s_endpgm)
body = lines[start:end + 1]
if "--dump" in sys.argv:
    lab = sys.argv[sys.argv.index("--dump") + 1]
    on = False
    for l in body:
        if re.match(r"^\.LBB\d+_\d+:", l):
            on = l.startswith(lab + ":")
        if on and not l.strip().startswith(";"):
            print(l)
    sys.exit(0)
blocks, cur = [], None
for ln in body:
    s = ln.strip()
    if not s or s.startswith(";") and not s.startswith("; %bb") or (s.startswith(".") and not s.startswith(".LBB")):
        continue
    m = re.match(r"^(\.LBB\d+_\d+|_Z\S+):", s)
    if m or s.startswith("; %bb"):
        cur = [m.group(1) if m else s.split()[1], [], ln]
        blocks.append(cur)
        continue
    cur[1].append(s.split()[0])


def cls(i):
    if i.startswith("v_"):
        return "VALU"
    if i.startswith("ds_"):
        return "LDS"
    if i.startswith(("global_", "flat_", "buffer_")):
        return "VMEM"
    if i.startswith(("s_waitcnt", "s_nop", "s_cbranch", "s_branch", "s_barrier", "s_endpgm")):
        return i[2:].split("_")[0]
    if i.startswith("s_load"):
        return "SMEM"
    return "SALU"


tot = 0
for nm, ins, ln in blocks:
    c = collections.Counter(cls(i) for i in ins)
    depth = re.search(r"Depth=(\d)", ln)
    print(f"{nm:12s} {len(ins):4d}  d={depth.group(1) if depth else '-'}  {dict(c)}")
    tot += len(ins)
print("total static", tot)
