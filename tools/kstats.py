#!/usr/bin/env python3
"""Print a rocprofv3 kernel_stats.csv as ms/step: python tools/kstats.py <csv> <steps>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1]))); steps = float(sys.argv[2])
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 24]:
    print(f"{float(r['Percentage']):6.2f}%  {float(r['TotalDurationNs'])/steps/1e6:8.3f} ms/step  avg {float(r['AverageNs'])/1e3:8.1f} us  n={r['Calls']:>5}  {r['Name'][:80]}")
print(f"total {tot/steps/1e6:.2f} ms/step of kernel time")
