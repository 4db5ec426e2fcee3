#!/usr/bin/env python3
"""GPU box: full-step student gradients on the goldens' inputs, HIP vs oracle (fp32 / bf16-emulating / with the HIP run's
ReLU gates injected): the table committed as profiles/r02_parity_emu.txt (tests/parity_helpers.py does the work).
usage: python tools/parity_emu.py [case ...]      cases: clip_small_2views_b16 clip_small_2views_b64 clip_small_6crops frame_small"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from parity_helpers import step_three_ways

for case in sys.argv[1:] or ["clip_small_2views_b16", "frame_small"]:
    losses, tab = step_three_ways(case)
    print(f"[{case}] " + "  ".join(f"{k} {v:.6f}" if isinstance(v, float) else f"{k} {v}" for k, v in losses.items()), flush=True)
    for k in ("HIP vs fp32", "emulated vs fp32", "HIP vs emulated", "fp32+gates vs fp32", "HIP vs fp32+gates", "HIP vs emulated+gates"):
        s = tab[k]
        print(f"  {k:24s} encoder mean {s['mean']:.3e} worst {s['worst'][0]:.3e} ({s['worst'][1]}) | "
              + "  ".join(f"{n} {v:.2e}" for n, v in s["heads"].items()), flush=True)
