#!/usr/bin/env python3
"""Idle time between consecutive kernels in a rocprofv3 kernel trace: python tools/gaps.py <kernel_trace.csv> [skip_first_n] [top]"""
import csv, re, sys
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
top = int(sys.argv[3]) if len(sys.argv) > 3 else 25
rows = rows[skip:]
short = lambda n: re.sub(r"\(anonymous namespace\)::|void |at::native::", "", n)[:60]
gaps = []
busy = 0
end = rows[0][1]
for i in range(1, len(rows)):
    s, e, n = rows[i]
    busy += rows[i - 1][1] - rows[i - 1][0]
    if s > end:
        gaps.append((s - end, short(rows[i - 1][2]), short(n)))
    end = max(end, e)
span = rows[-1][1] - rows[0][0]
tot = sum(g[0] for g in gaps)
print(f"span {span / 1e6:.2f} ms, kernels {len(rows)}, idle {tot / 1e6:.2f} ms ({100 * tot / span:.1f} %), gaps >5us: {sum(1 for g in gaps if g[0] > 5000)}")
import collections
agg = collections.defaultdict(lambda: [0, 0])
for g, a, b in gaps:
    agg[(a, b)][0] += g; agg[(a, b)][1] += 1
for (a, b), (g, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:top]:
    print(f"{g / 1e3:9.1f} us total  x{c:4d}  avg {g / c / 1e3:7.1f} us   {a}  ->  {b}")
