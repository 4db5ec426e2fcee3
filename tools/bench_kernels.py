#!/usr/bin/env python3
"""Run bench.py (clip2) with optional gemm tuning-hook values and print the per-kernel table: python tools/bench_kernels.py [hooks...]"""
import json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
env = dict(os.environ, ATST_TUNE=",".join(sys.argv[1:]))
out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", "clip2", "--steps", "6", "--warmup", "2", "--no-cpu-baseline"],
                     capture_output=True, text=True, env=env).stdout.strip().splitlines()[-1]
d = json.loads(out)
print("hooks", sys.argv[1:], "->", d["value"], "clips/s", d["ms_per_step"], "ms/step")
for k in d["kernels"]:
    print(f"   {k['kernel']:32s} n={k['launches']:4d} avg {k['avg_us']:7.1f} us  total {k['total_ms']/6:7.3f} ms/step  {k['achieved']:8.1f} {k['unit']}")
