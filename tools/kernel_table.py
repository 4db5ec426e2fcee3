#!/usr/bin/env python3
"""Per-(kernel, grid) table from a rocprofv3 --kernel-trace CSV: launches per step, average us, ms per step.
usage: python tools/kernel_table.py <dir-or-csv> <steps> [out.txt]       (run after: rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 bench.py ...)"""
import csv, glob, os, re, sys
from collections import defaultdict
src, steps = sys.argv[1], int(sys.argv[2])
files = [src] if os.path.isfile(src) else glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)
agg = defaultdict(lambda: [0, 0.0])
for f in files:
    with open(f) as fh:
        for r in csv.DictReader(fh):
            name = r["Kernel_Name"]
            name = re.sub(r"\(anonymous namespace\)::", "", name)
            name = re.sub(r"^void ", "", name)
            name = re.sub(r"\((?:[^()]|\([^()]*\))*\)$", "", name)
            if len(name) > 70: name = name[:67] + "..."
            grid = int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1)
            k = (name, grid)
            agg[k][0] += 1
            agg[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
tot = sum(v[1] for v in agg.values())
lines = [f"# {tot / steps / 1e3:.3f} ms of kernel time per step over {steps} steps ({len(files)} trace file(s))",
         f"{'kernel':70s} {'blocks':>7s} {'n/step':>7s} {'avg us':>9s} {'ms/step':>8s} {'share':>6s}"]
for (name, grid), (n, us) in rows:
    if us / tot < 0.0005: continue
    lines.append(f"{name:70s} {grid:7d} {n / steps:7.2f} {us / n:9.1f} {us / steps / 1e3:8.3f} {100 * us / tot:5.1f}%")
out = "\n".join(lines)
print(out)
if len(sys.argv) > 3:
    open(sys.argv[3], "w").write(out + "\n")
