#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output: python tools/pmc_summary.py <dir> [kernel-substring]"""
import collections, csv, glob, sys
d = sys.argv[1]; sub = sys.argv[2] if len(sys.argv) > 2 else ""
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-60:]
        if sub in r["Kernel_Name"]:
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
    for k, v in agg.items():
        print(k)
        for c, x in sorted(v.items()):
            print(f"   {c:32s} {x / cnt[(k, c)]:16.1f}  (n={cnt[(k, c)]})")
for f in glob.glob(d + "/**/*kernel_stats.csv", recursive=True):
    print(open(f).read()[:6000])
