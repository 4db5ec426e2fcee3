#!/usr/bin/env python3
"""Compact view of a bench.py JSON line (stdin): headline, roofline, per-kernel table."""
import json, sys
for line in sys.stdin:
    line = line.strip()
    if not line.startswith("{"):
        continue
    d = json.loads(line)
    print(f"{d['config']['workload'][:40]:40s} {d['value']:9.2f} {d['unit']}  {d['ms_per_step']:8.3f} ms/step  n_gpus {d['n_gpus']}  step-MFMA-frac {d.get('mfma_roofline_frac_step')}")
    r = d.get("roofline")
    if r:
        print("  roofline:", {k: r[k] for k in ("kernel", "bound", "achieved", "unit", "frac", "traffic", "avg_launch_us") if k in r})
    for k in d.get("kernels", []):
        print(f"   {k['kernel']:30s} n={k['launches']:5d} avg {k['avg_us']:8.1f} us  total {k['total_ms']:8.2f} ms  {k['bound']:4s} {k['achieved']:8.1f} {k['unit']}" + (f"  ({k['tflops']} TF/s)" if "tflops" in k else ""))
    if "cpu_baseline" in d:
        print("  cpu_baseline:", d["cpu_baseline"]["value"], d["cpu_baseline"]["unit"], "cores", d["cpu_baseline"]["cores"])
