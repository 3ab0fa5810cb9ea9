#!/usr/bin/env python3
"""Print per-kernel PMC counter averages from a rocprofv3 -i <file> output directory."""
import collections, csv, glob, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(sys.argv[1] + '/pmc_*/*counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].replace('mgpu::', '').replace('void ', '')
        if k.startswith('__amd'): continue
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"   {c:28s} {sum(v)/len(v):16.0f}   (n={len(v)})")
