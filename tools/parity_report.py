#!/usr/bin/env python3
"""Print HIP-vs-reference-golden error tables (run on the GPU box): python tools/parity_report.py"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from audiossl_amd.engine import AtstEngine
from audiossl_amd import hip
from oracle import atst_oracle as O

def load(n): return np.load(os.path.join(ROOT, "tests", "golden", n + ".npz"))
def sidx(n, k=192): return np.unique(np.linspace(0, n - 1, num=min(n, k)).astype(np.int64))
def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))

def table(eng, G, only_encoder=False, strip=""):
    rows = []
    for name, (off, shape) in eng.layout.entries.items():
        key = name[len(strip):] if strip and name.startswith(strip) else name
        if only_encoder and not name.startswith("encoder."): continue
        if "gsamp/" + key not in G: continue
        g = eng.param_view("student", name, grad=True).reshape(-1).double().cpu()
        r = rel(g[sidx(g.numel())].numpy(), G["gsamp/" + key]); gn = float(G["gnorm/" + key])
        rows.append((name, r, float(g.norm()) / gn - 1, g.numel()))
    tot = sum(r[3] for r in rows)
    print(f"   weighted mean rel {sum(r[1]*r[3] for r in rows)/tot:.3e}  max {max(r[1] for r in rows):.3e}")
    for r in rows:
        if any(t in r[0] for t in ("cls_token", "pos_embed", "patch_embed.patch_embed.weight", "blocks.0.attn.qkv", "blocks.6.attn.qkv", "blocks.11.mlp.fc2.weight", "norm.weight", "projector", "predictor", "mask_embed")):
            print(f"   {r[0]:45s} rel {r[1]:.3e}  norm {r[2]:+.2e}")

def clip(name):
    G = load(name); B, ncrops = int(G["B"]), int(G["ncrops"]); widths = [int(w) for w in G["widths"]]
    drop = "keep_t0" in G
    eng = AtstEngine("small", ncrops=ncrops, drop_path_rate=0.1 if drop else 0.0)
    eng.load_weights(O.recipe_weights("small", seed=int(G["seed_w"])))
    mels = [O.recipe_mel(B, w, seed=int(G["seed_x"]) + i) for i, w in enumerate(widths)]
    lens = [torch.from_numpy(l) for l in G["lengths"]]
    kt = ks = None
    if drop:
        kt = [torch.from_numpy(G[f"keep_t{i}"]) for i in range(len(O.group_views(widths[:2])))]
        ks = [torch.from_numpy(G[f"keep_s{i}"]) for i in range(len(O.group_views(widths)))]
    loss, ss, st = eng.forward(mels, lens, None, kt, ks); eng.backward()
    so, to = eng.last_outputs
    print(f"[{name}] loss {loss.item():.6f}/{float(G['loss']):.6f} std_s {ss.item():.5f}/{float(G['std_s']):.5f} out rel {rel(so.cpu().numpy()[:8], G['student_out']):.2e} {rel(to.cpu().numpy()[:8], G['teacher_out']):.2e}")
    table(eng, G)

def encgrad():
    G = load("clip_encoder_grad"); S = int(G["S"])
    eng = AtstEngine("small"); eng.load_weights(O.recipe_weights("small", seed=21))
    ep = eng._pass("student", S, 1001, True, 0)
    out = ep.forward(O.recipe_mel(S, 1001, seed=23).cuda(), eng._valid(torch.from_numpy(G["length"]), 1), None,
                     eng.drop_path_scales(S, torch.from_numpy(G["keep"])))
    cls = out.float().reshape(S, 256, 384)[:, 0]
    print(f"[encoder_grad] cls rel {rel(cls.cpu().numpy(), G['cls']):.3e}")
    R = torch.from_numpy(np.random.default_rng(29).standard_normal((S, 384)).astype(np.float32)).cuda()
    eng.g32.zero_(); ep.dout.zero_()
    rows = (torch.arange(S, dtype=torch.int32, device="cuda") * 256).contiguous()
    hip.call("atst_scatter_rows_bf16", hip.ptr(R), hip.ptr(rows), S, 384, hip.ptr(ep.dout), hip.stream())
    ep.backward()
    table(eng, G, only_encoder=True, strip="encoder.")

if __name__ == "__main__":
    encgrad()
    for n in ("clip_small_2views_b64", "clip_small_2views_b16", "clip_small_2views", "clip_small_6crops"):
        clip(n)
