#!/usr/bin/env python3
"""Per-step kernel time split by (kernel, grid size) from a rocprofv3 --kernel-trace CSV of `bench.py --steps K --warmup W` (the launches of the
local views and of the global views of one kernel differ only in their grid).  usage: python tools/trace_by_grid.py t_kernel_trace.csv <K + W>"""
import csv, re, sys
from collections import defaultdict
rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2])
d = defaultdict(lambda: [0, 0])
for r in rows:
    n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]); n = re.sub(r"^void ", "", n); n = re.sub(r"\(.*", "", n)[:48]
    k = (n, int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]))
    d[k][0] += 1; d[k][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
tot = sum(v[1] for v in d.values())
print("kernel time per step %.2f ms" % (tot / steps / 1e6))
for (n, g), (c, t) in sorted(d.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[3]) if len(sys.argv) > 3 else 45]:
    print("%-50s blocks %6d  n/step %5.1f  avg %8.1f us  %6.3f ms/step" % (n, g, c / steps, t / c / 1e3, t / steps / 1e6))
