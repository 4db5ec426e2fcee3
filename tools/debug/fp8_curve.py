#!/usr/bin/env python3
"""Loss curves of N optimizer steps on the same data and initial weights: bf16 | e4m3 forward only | all 12 GEMMs of a block on e4m3 operands (GPU box;
env N, DEPTH, B: e.g. DEPTH=12 N=48 B=32 for the full-depth ATST-base curve of profiles/r05_fp8_curve_depth12.txt).
Prints per-step losses and the parameter distance to the bf16 run; tests/test_ops_gpu.py::test_fp8_multi_step_loss_curve_tracks_bf16 pins it."""
import math, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from audiossl_amd.engine import AtstEngine
from oracle import atst_oracle as O
N, depth, B = int(os.environ.get("N", 24)), int(os.environ.get("DEPTH", 2)), int(os.environ.get("B", 16))
ARCH, CROPS = os.environ.get("ARCH", "base"), int(os.environ.get("CROPS", 2))     # round 6: ARCH=small CROPS=6 = the clip6 recipe (2 views of 10 s + 4 of 1 s) at d = 384
W = O.recipe_weights(ARCH, depth=depth, seed=7)


def data(step):
    g = torch.Generator().manual_seed(1000 + step)
    mels = []
    for v in range(2):
        m = O.recipe_mel(B, 1001, seed=10 * step + v)
        # structure: a few horizontal ridges (tones) and a slow envelope, different per clip -- the head rows must not be near-identical
        f = torch.rand(B, 1, 64, 1, generator=g); tt = torch.linspace(0, 1, 1001).view(1, 1, 1, -1)
        m = m * 0.3 + 0.7 * torch.sin(6.28 * (3 * f + 2 * tt * torch.rand(B, 1, 1, 1, generator=g)))
        mels.append(m.contiguous())
    for v in range(CROPS - 2):                                   # 1 s local views: a window of the first global view
        o = int(torch.randint(0, 900, (1,), generator=g))
        mels.append(mels[0][..., o:o + 101].contiguous())
    return mels


def run(kind):
    eng = AtstEngine(ARCH, depth=depth, ncrops=CROPS, drop_path_rate=0.0, fp8=kind != "bf16")
    eng.load_weights(W)
    if kind == "fp8_fwd":
        eng.fp8_bwd_state = 0
    losses = []
    for step in range(N):
        mels = [m.cuda() for m in data(step)]
        lens = [torch.full((B,), 1001)] * 2 + [torch.full((B,), 101)] * (CROPS - 2)
        loss = eng.forward(mels, lens)[0]
        eng.backward()
        eng.optimizer_step(5e-4, 0.04, 0.99)
        losses.append(float(loss))
    sat = eng.fp8_saturation() if kind != "bf16" else {}
    return losses, eng.p32.clone(), sat


_e0 = AtstEngine(ARCH, depth=depth, ncrops=CROPS, drop_path_rate=0.0); _e0.load_weights(W); P0 = _e0.p32.clone(); del _e0
ref, p_ref, _ = run("bf16")
print("step  " + " ".join(f"{i:7d}" for i in range(N)))
print("bf16  " + " ".join(f"{v:7.4f}" for v in ref))
for kind in ("fp8_fwd", "fp8"):
    l, p, sat = run(kind)
    d = [abs(a - b) for a, b in zip(l, ref)]
    print(f"{kind:7s}" + " ".join(f"{v:7.4f}" for v in l))
    print(f"      max |loss - bf16| {max(d):.4f}  mean {sum(d) / N:.4f}  last-4 mean loss {sum(l[-4:]) / 4:.4f} (bf16 {sum(ref[-4:]) / 4:.4f})  "
          f"|p - p_bf16| / |p_bf16 - p_0| {float((p - p_ref).norm() / (p_ref - P0).norm()):.3f}  sat {sat}")
