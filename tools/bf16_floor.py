#!/usr/bin/env python3
"""CPU-only: how far do the student gradients of the fp32 algorithm move under bf16 rounding alone?

  (A) inputs and weights rounded to bf16 once, everything else fp32   -- the smallest perturbation any bf16 path makes
  (B) oracle.emulate_bf16(): rounding at every point where the HIP path rounds (GEMM operands, saved activations, gradient
      operands), fp32 elsewhere                                        -- the floor of THIS design
both against the plain fp32 oracle (pinned to the reference by tests/test_oracle_golden.py), full training step
(teacher + student + BatchNorm/ReLU heads + BYOL loss), ATST-small, 12 layers, 2 views of 10 s, DropPath off.
Writes the table to stdout (committed as profiles/r02_bf16_floor.txt).   usage: python tools/bf16_floor.py [B ...]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import atst_oracle as O

torch.set_num_threads(max(1, (os.cpu_count() or 2)))


def grads(W, mels, lens, emu):
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in W.items() if k.startswith("student.") and v.dtype == torch.float32
              and "running" not in k}
    Wl = {k: (leaves[k] if k in leaves else v.clone()) for k, v in W.items()}
    with O.emulate_bf16(emu):
        loss, _, _ = O.atst_forward(Wl, mels, lens, "small", 2, drop_path_rate=0.0)
        loss.backward()
    return float(loss), {k[len("student."):]: v.grad.detach() for k, v in leaves.items() if v.grad is not None}


def rel(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))


def table(tag, g, ref):
    enc = [(k, rel(g[k], ref[k]), ref[k].numel()) for k in ref if k.startswith("encoder.") and k in g]
    w = sum(n for _, _, n in enc)
    mean = sum(r * n for _, r, n in enc) / w
    worst = max(enc, key=lambda t: t[1])
    heads = {k: rel(g[k], ref[k]) for k in ("projector.0.weight", "projector.3.weight", "predictor.0.weight", "predictor.3.weight")}
    print(f"  {tag:34s} encoder: param-weighted mean {mean:.3e}  worst {worst[1]:.3e} ({worst[0]})  | "
          + "  ".join(f"{k} {v:.2e}" for k, v in heads.items()))


for B in [int(a) for a in sys.argv[1:]] or [2, 8]:
    W = O.recipe_weights("small", seed=0)
    mels = [O.recipe_mel(B, 1001, seed=21), O.recipe_mel(B, 1001, seed=22)]
    lens = [torch.full((B,), 1001)] * 2
    l0, g0 = grads(W, mels, lens, False)
    Wr = {k: (O._bf(v) if v.dtype == torch.float32 and "running" not in k else v) for k, v in W.items()}
    l1, g1 = grads(Wr, [O._bf(m) for m in mels], lens, False)
    l2, g2 = grads(W, mels, lens, True)
    print(f"B = {B} clips x 2 views ({2 * B} BatchNorm rows)   loss fp32 {l0:.6f} | (A) {l1:.6f} | (B) {l2:.6f}")
    table("(A) bf16 inputs + weights vs fp32", g1, g0)
    table("(B) full bf16 emulation vs fp32", g2, g0)
