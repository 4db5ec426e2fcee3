#!/usr/bin/env python3
"""Summarise rocprofv3 counter CSVs per kernel symbol: python tools/pmc_summary.py <dir> [kernel-substring]"""
import collections, csv, glob, re, sys
d = sys.argv[1]; sub = sys.argv[2] if len(sys.argv) > 2 else ""
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).replace("void ", "")
        name = re.sub(r"\((GemmArgs|WgradArgs|AttnArgs|LnBwdArgs|OptimArgs).*", "", name)[:70]
        if sub in name:
            agg[name][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(name, r["Counter_Name"])] += 1
    for k, v in sorted(agg.items()):
        for c, x in sorted(v.items()):
            print(f"{k:60s} {c:14s} avg/launch {x / cnt[(k, c)]:14.1f}  launches {cnt[(k, c)]}")
