#!/usr/bin/env python3
"""Per-kernel-kind comparison of two bench.py JSON lines (in-library HIP-event timing): python tools/ab_kinds.py old.json new.json"""
import json, sys
a, b = (json.load(open(f)) for f in sys.argv[1:3])
ka = {k["kernel"]: k for k in a["kernels"]}; kb = {k["kernel"]: k for k in b["kernels"]}
print(f"step: {a['ms_per_step']:.2f} -> {b['ms_per_step']:.2f} ms")
for name in sorted(set(ka) | set(kb), key=lambda n: -(ka.get(n, kb.get(n))["total_ms"])):
    x, y = ka.get(name), kb.get(name)
    f = lambda k: f"n={k['launches']:5d} avg {k['avg_us']:8.1f} us total {k['total_ms']:8.2f} ms" if k else " " * 45
    print(f"{name:30s} {f(x)}  |  {f(y)}")
