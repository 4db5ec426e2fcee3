#!/usr/bin/env python3
"""Where the bf16 path's 6.9e-3 gradient error comes from.  CPU only (build container or GPU box host): the oracle's bf16 emulation
(oracle.atst_oracle.emulate_bf16: rounds exactly where the HIP path rounds) against the REFERENCE golden of the smooth encoder-only objective
(tests/golden/clip_encoder_grad.npz: 12 layers, ragged lengths, DropPath), with the rounding sites switched off one at a time, then in groups.

    python tools/rounding_sites.py > profiles/r05_rounding_sites.txt

Columns: parameter-weighted mean / worst tensor of the per-tensor rel-L2 gradient error vs the golden; `d mean` = what leaving that site in fp32 buys."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import atst_oracle as O
torch.set_num_threads(min(os.cpu_count() or 1, 16))

G = np.load(os.path.join(ROOT, "tests", "golden", "clip_encoder_grad.npz"))
S = int(G["S"])
R = torch.from_numpy(np.random.default_rng(29).standard_normal((S, 384)).astype(np.float32))
mel, length, keep = O.recipe_mel(S, 1001, seed=23), torch.from_numpy(G["length"]), torch.from_numpy(G["keep"])


def sidx(n, k=192):
    return np.unique(np.linspace(0, n - 1, num=min(n, k)).astype(np.int64))


def run(off=(), on=True):
    W = O.recipe_weights("small", seed=21)
    leaves = {k[len("student.encoder."):]: v.requires_grad_(True) for k, v in W.items() if k.startswith("student.encoder.")}
    with O.emulate_bf16(on, off=off):
        cls = O.encoder_forward(W, "student.encoder.", mel, length, "small", keep=keep)
        (cls * R).sum().backward()
    num = den = 0.0
    worst = (0.0, "")
    for name, p in leaves.items():
        if "gsamp/" + name not in G:
            continue
        g = p.grad.reshape(-1).double()
        ref = G["gsamp/" + name].astype(np.float64)
        r = float(np.linalg.norm(g[sidx(g.numel())].numpy() - ref) / (np.linalg.norm(ref) + 1e-30))
        num += r * g.numel(); den += g.numel()
        worst = max(worst, (r, name))
    cls_err = float(np.linalg.norm(cls.detach().numpy() - G["cls"]) / np.linalg.norm(G["cls"]))
    return num / den, worst, cls_err


t0 = time.time()
fp32 = run(on=False)
base = run()
print(f"reference golden: clip_encoder_grad (S = {S}, 12 layers, ragged lengths, DropPath); errors are per-tensor rel-L2 of the encoder gradient on 192 samples per tensor")
print(f"{'configuration':44s} {'mean':>10s} {'worst tensor':>12s}  {'CLS rel-L2':>10s}  {'d mean':>8s}")
print(f"{'fp32 oracle (no rounding)':44s} {fp32[0]:10.3e} {fp32[1][0]:12.3e}  {fp32[2]:10.3e}")
print(f"{'bf16 emulation, every site on':44s} {base[0]:10.3e} {base[1][0]:12.3e}  {base[2]:10.3e}   (worst: {base[1][1]})")
rows = []
for site in O.ROUNDING_SITES:
    m, w, c = run(off=(site,))
    rows.append((base[0] - m, site, m, w, c))
    print(f"{'  without ' + site:44s} {m:10.3e} {w[0]:12.3e}  {c:10.3e}  {base[0] - m:+8.1e}", flush=True)
groups = {"all forward values (w .. ln_final)": [s for s in O.ROUNDING_SITES if not s.startswith("g_")],
          "all gradient operands (g_*)": [s for s in O.ROUNDING_SITES if s.startswith("g_")],
          "weights + LayerNorm outputs (w, ln1, ln2, ln_final)": ["w", "ln1", "ln2", "ln_final"],
          "attention internals (qkv, P, attn_out, g_qkv, g_S, g_attn_out)": ["qkv", "P", "attn_out", "g_qkv", "g_S", "g_attn_out"],
          "MLP internals (gelu_out, u_saved, g_fc1_out)": ["gelu_out", "u_saved", "g_fc1_out"],
          "branch gradients (g_proj_out, g_fc2_out)": ["g_proj_out", "g_fc2_out"],
          "the five largest single sites": [r[1] for r in sorted(rows, reverse=True)[:5]]}
print()
for label, sites in groups.items():
    m, w, c = run(off=tuple(sites))
    print(f"{'  without ' + label:72s} {m:10.3e} {w[0]:12.3e}  {c:10.3e}  {base[0] - m:+8.1e}", flush=True)
print(f"\n({time.time() - t0:.0f} s on {torch.get_num_threads()} threads)")
