#!/usr/bin/env python3
"""Per-kernel difference of two rocprofv3 *_kernel_stats.csv of the same bench command (old tree, new tree; same box, same call):
python tools/ab_trace.py old_kernel_stats.csv new_kernel_stats.csv <steps incl. warm-up>"""
import csv, re, sys
def load(f):
    d = {}
    for r in csv.DictReader(open(f)):
        n = re.sub(r"\(anonymous namespace\)::", "", r["Name"]); n = re.sub(r"^void ", "", n); n = re.sub(r"\(.*", "", n)[:60]
        c, t = d.get(n, (0, 0)); d[n] = (c + int(r["Calls"]), t + int(r["TotalDurationNs"]))
    return d
o, n, steps = load(sys.argv[1]), load(sys.argv[2]), int(sys.argv[3])
print("kernel time per step: old %.2f ms, new %.2f ms" % (sum(v[1] for v in o.values()) / steps / 1e6, sum(v[1] for v in n.values()) / steps / 1e6))
rows = [(n.get(k, (0, 0))[1] - o.get(k, (0, 0))[1], k, o.get(k, (0, 0)), n.get(k, (0, 0))) for k in set(o) | set(n)]
for d, k, a, b in sorted(rows, key=lambda r: -abs(r[0]))[:int(sys.argv[4]) if len(sys.argv) > 4 else 16]:
    print("%+7.3f ms/step  %-60s old n=%5.1f %8.1f us | new n=%5.1f %8.1f us" % (d / steps / 1e6, k, a[0] / steps, a[1] / max(a[0], 1) / 1e3, b[0] / steps, b[1] / max(b[0], 1) / 1e3))
