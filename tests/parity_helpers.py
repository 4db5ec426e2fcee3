"""Shared by tests/test_step_gpu.py and tools/parity_emu.py (NOT a test module): one full training step of the goldens'
inputs evaluated by the HIP path, by the fp32 oracle and by the oracle's bf16-emulation mode, optionally with the HIP
run's ReLU gate patterns injected into the oracle (oracle.relu_gates), and per-tensor rel-L2 summaries of the student
gradients."""
import os
import numpy as np
import torch
from audiossl_amd.engine import AtstEngine
from oracle import atst_oracle as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CANCELLING = ("encoder.pos_embed", "encoder.norm.bias", "encoder.norm_frame.bias")     # sums that cancel across the batch behind BatchNorm
HEADS = ("projector.0.weight", "projector.3.weight", "predictor.0.weight", "predictor.3.weight")


def rel(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))


def oracle_grads(W, fwd, emu, gates=None, ctxs=()):
    """ctxs: further oracle contexts entered inside emulate_bf16 (emulate_fp8, emulate_fp8_dgrad)"""
    import contextlib
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in W.items() if k.startswith("student.") and v.dtype == torch.float32
              and "running" not in k}
    Wl = {k: (leaves[k] if k in leaves else v.clone()) for k, v in W.items()}
    with O.emulate_bf16(emu), O.relu_gates(gates or {}), contextlib.ExitStack() as stack:
        for c in ctxs:
            stack.enter_context(c)
        loss = fwd(Wl)[0]
        loss.backward()
    return float(loss.detach()), {k[len("student."):]: v.grad.detach() for k, v in leaves.items() if v.grad is not None}


def summarize(g, ref):
    """-> dict(mean=parameter-weighted mean over encoder tensors, worst=(value, name), heads={name: rel})."""
    enc = [(k, rel(g[k], ref[k]), ref[k].numel()) for k in ref if k.startswith("encoder.") and k in g and k not in CANCELLING]
    mean = sum(r * n for _, r, n in enc) / sum(n for _, _, n in enc)
    worst = max(enc, key=lambda t: t[1])
    return dict(mean=mean, worst=(worst[1], worst[0]), heads={k: rel(g[k], ref[k]) for k in HEADS})


def step_three_ways(name, with_plain=True):
    """-> (losses dict, {comparison: summary}) for golden case `name`."""
    G = np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False)
    frame = name.startswith("frame")
    B = int(G["B"])
    if frame:
        W = O.recipe_weights("small", frame=True, seed=11)
        mels = [O.recipe_mel(B, 1001, seed=21), O.recipe_mel(B, 1001, seed=22)]
        lens = [torch.from_numpy(l) for l in G["lengths"]]
        masks = [torch.from_numpy(G["mask"])] * 2
        kt, ks = [torch.from_numpy(G["keep_t0"])], [torch.from_numpy(G["keep_s0"])]
        eng = AtstEngine("small", frame=True)
        fwd = lambda Wl: O.frame_atst_forward(Wl, mels, lens, masks, "small", kt, ks)
    else:
        ncrops = int(G["ncrops"]); widths = [int(w) for w in G["widths"]]
        drop = "keep_t0" in G
        W = O.recipe_weights("small", seed=int(G["seed_w"]))
        mels = [O.recipe_mel(B, w, seed=int(G["seed_x"]) + i) for i, w in enumerate(widths)]
        lens = [torch.from_numpy(l) for l in G["lengths"]]
        masks, kt, ks = None, None, None
        if drop:
            kt = [torch.from_numpy(G[f"keep_t{i}"]) for i in range(len(O.group_views(widths[:2])))]
            ks = [torch.from_numpy(G[f"keep_s{i}"]) for i in range(len(O.group_views(widths)))]
        eng = AtstEngine("small", ncrops=ncrops, drop_path_rate=0.1 if drop else 0.0)
        fwd = lambda Wl: O.atst_forward(Wl, mels, lens, "small", ncrops, kt, ks, drop_path_rate=0.1 if drop else 0.0)
    eng.load_weights(W)
    loss_h, _, _ = eng.forward(mels, lens, masks, kt, ks)
    # ReLU gate patterns of the HIP run: sign of the saved BatchNorm+ReLU output (columns [0, 4096) of the split operand)
    gates = {f"student.{w}.": (eng.heads[f"student.{w}"].saved[4][:, :4096].float() > 0).cpu() for w in ("projector", "predictor")}
    eng.backward()
    gh = {k: eng.param_view("student", k, grad=True).detach().cpu().clone() for k in eng.layout.entries}
    l32, g32 = oracle_grads(W, fwd, False)
    gh = {k: v for k, v in gh.items() if k in g32}
    _, g32g = oracle_grads(W, fwd, False, gates)
    lem_g, gemg = oracle_grads(W, fwd, True, gates)
    losses = dict(hip=float(loss_h), fp32=l32, golden=float(G["loss"]), B=B)
    out = {"HIP vs fp32": summarize(gh, g32), "HIP vs fp32+gates": summarize(gh, g32g), "HIP vs emulated+gates": summarize(gh, gemg),
           "fp32+gates vs fp32": summarize(g32g, g32)}
    if with_plain:
        lem, gem = oracle_grads(W, fwd, True)
        losses["emulated"] = lem
        out["emulated vs fp32"] = summarize(gem, g32)
        out["HIP vs emulated"] = summarize(gh, gem)
    return losses, out
