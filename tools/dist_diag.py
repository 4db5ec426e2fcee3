#!/usr/bin/env python3
"""Diagnostic for tests/test_dist_gpu.py: gradient differences (per flat-buffer section) between the single-process run,
a second single-process run (run-to-run noise floor) and the 2-rank runs with the bucketed overlap off / on."""
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from audiossl_amd.engine import FlatLayout
mode = sys.argv[1] if len(sys.argv) > 1 else "clip"
batch = sys.argv[2] if len(sys.argv) > 2 else "4"
def run(tag, world, overlap=1):
    out = f"/tmp/dd_{tag}.npz"
    base = [os.path.join(ROOT, "tests", "dist_worker.py"), "--mode", mode, "--out", out, "--overlap", str(overlap), "--batch", batch]
    cmd = [sys.executable] + base if world == 1 else [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
                                                      "--master-addr", "127.0.0.1", "--master-port", "29577"] + base
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return np.load(out)
L = FlatLayout("small", 4, mode != "clip")
def rel(a, b): return float(np.linalg.norm(a.astype(np.float64) - b) / (np.linalg.norm(b.astype(np.float64)) + 1e-30))
one, one2, two0, two1 = run("a", 1), run("b", 1), run("c", 2, 0), run("d", 2, 1)
print("loss", one["loss"], one2["loss"], two0["loss"], two1["loss"])
secs = {"patch+tokens": (0, L.entries["encoder.blocks.0.norm1.weight"][0])}
for i in range(4):
    a = L.entries[f"encoder.blocks.{i}.norm1.weight"][0]
    b = L.entries[f"encoder.blocks.{i+1}.norm1.weight"][0] if i < 3 else L.entries["projector.0.weight"][0]
    secs[f"block{i}"] = (a, b)
secs["projector"] = (L.entries["projector.0.weight"][0], L.entries["predictor.0.weight"][0])
secs["predictor"] = (L.entries["predictor.0.weight"][0], L.n_student)
print(f"{'section':14s} {'1 vs 1 (noise)':>16s} {'2 ranks ovl=0':>16s} {'2 ranks ovl=1':>16s}")
for k, (a, b) in secs.items():
    print(f"{k:14s} {rel(one2['grads'][a:b], one['grads'][a:b]):16.3e} {rel(two0['grads'][a:b], one['grads'][a:b]):16.3e} {rel(two1['grads'][a:b], one['grads'][a:b]):16.3e}")
print(f"{'all':14s} {rel(one2['grads'], one['grads']):16.3e} {rel(two0['grads'], one['grads']):16.3e} {rel(two1['grads'], one['grads']):16.3e}")
