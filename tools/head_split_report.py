#!/usr/bin/env python3
"""Deviation of AtstEngine(head_split2="gated") -- the default since round 5: the second Linear of the teacher projector and of the student predictor on plain
bf16 operands -- from head_split2="all" (rounds 1-4: every second head Linear in split-bf16, ~2^-16), against the reference goldens.
Run on the GPU box: python tools/head_split_report.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from audiossl_amd.engine import AtstEngine
from oracle import atst_oracle as O
import test_step_gpu as T

def run(name, mode):
    G = T.load(name)
    if name.startswith("frame"):
        B = int(G["B"])
        eng = AtstEngine("small", frame=True, head_split2=mode)
        eng.load_weights(O.recipe_weights("small", frame=True, seed=11))
        mels = [O.recipe_mel(B, 1001, seed=21), O.recipe_mel(B, 1001, seed=22)]
        lens = [torch.from_numpy(l) for l in G["lengths"]]; masks = [torch.from_numpy(G["mask"])] * 2
        loss, _, _ = eng.forward(mels, lens, masks, [torch.from_numpy(G["keep_t0"])], [torch.from_numpy(G["keep_s0"])])
        so, to = None, None
    else:
        B, ncrops = int(G["B"]), int(G["ncrops"]); widths = [int(w) for w in G["widths"]]
        eng = AtstEngine("small", ncrops=ncrops, drop_path_rate=0.1, head_split2=mode)
        eng.load_weights(O.recipe_weights("small", seed=int(G["seed_w"])))
        mels = [O.recipe_mel(B, w, seed=int(G["seed_x"]) + i) for i, w in enumerate(widths)]
        lens = [torch.from_numpy(l) for l in G["lengths"]]
        kt = [torch.from_numpy(G[f"keep_t{i}"]) for i in range(len(O.group_views(widths[:2])))]
        ks = [torch.from_numpy(G[f"keep_s{i}"]) for i in range(len(O.group_views(widths)))]
        loss, _, _ = eng.forward(mels, lens, None, kt, ks)
    eng.backward()
    s_out, t_out = eng.last_outputs
    tab = T.grad_table(eng, G)
    keep = {k: v for k, v in tab.items() if k not in T.CANCELLING}
    mean = sum(r * n for r, _, n in keep.values()) / sum(n for _, _, n in keep.values())
    return float(loss), float(G["loss"]), s_out.float().cpu(), t_out.float().cpu(), mean, eng.g32.clone()

for name in ("clip_small_2views_b16", "clip_small_6crops", "frame_small"):
    a = run(name, "all"); g = run(name, "gated")
    rel = lambda x, y: float((x - y).norm() / y.norm())
    print(f"{name:24s} loss all {a[0]:.6f} gated {g[0]:.6f} (reference {a[1]:.6f}) | student out gated vs all {rel(g[2], a[2]):.2e}, teacher out {rel(g[3], a[3]):.2e} | "
          f"gradient vs golden: all {a[4]:.3e} gated {g[4]:.3e} | flat gradient gated vs all {rel(g[5], a[5]):.2e}")
