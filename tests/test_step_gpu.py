"""End-to-end parity of the HIP training step (bf16 MFMA compute; fp32 statistics, residual stream, master weights)
against goldens produced by the upstream reference (tests/golden/make_golden.py) and against the CPU oracle, through
the C ABI.  The reference is fp32; the HIP path rounds GEMM operands and saved activations to bf16 (2^-9 per element).

Tolerances (each justified by a measurement recorded in DESIGN.md "Precision"):
  * integer / index work (patch gather, valid lengths, ragged row order, untouched mask_embed): bit-exact
  * forward quantities: tokens 5e-3, block outputs / CLS 1e-2 rel-L2, head outputs 2.5e-2, loss |d| 5e-3, std 2e-3
  * gradients on a smooth objective (encoder-only golden, 12 layers, ragged lengths, DropPath): per tensor 2.4e-2,
    parameter-weighted mean 1e-2                                     (measured: mean 6.9e-3, max 1.57e-2; x1.5)
  * head + loss backward given identical input features (oracle run on the HIP features): 1.5e-2
  * full step vs the fp32 reference: gradients pass through two BatchNorm+ReLU heads whose gates are discontinuous; a
    1 % forward perturbation flips ~1 % of the gates and moves every upstream gradient by ~sqrt(1 %) = 10 %
    (the fp32 reference itself moves by 10-27 % when its inputs are merely rounded to bf16).  Checked: last-layer
    gradients (no gate behind them) 3e-2, every tensor's norm within 5 % (B >= 16), per tensor rel-L2 <= 0.19 (B >= 16).
    test_gradient_gap_is_the_relu_gates pins that gap: with the HIP run's gate patterns injected into the fp32 oracle the
    difference drops to 1.3-1.7 % (0.4-1.1 % against the bf16-emulating oracle), and the gate patterns alone move the
    fp32 oracle by the whole gap (profiles/r02_parity_emu.txt, profiles/r02_bf16_floor.txt).
  north_star's "student grads within 1e-3 rel" is therefore NOT met by the bf16 path (nor could any bf16 path meet it).
"""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from audiossl_amd import hip  # noqa: E402
from audiossl_amd.engine import AtstEngine  # noqa: E402
from oracle import atst_oracle as O  # noqa: E402

GOLD = os.path.join(os.path.dirname(__file__), "golden")
CANCELLING = ("encoder.pos_embed", "encoder.norm.bias", "encoder.norm_frame.bias")     # sums that cancel across the batch behind BatchNorm


def load(name):
    return np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False)


def sample_idx(n, k=192):
    return np.unique(np.linspace(0, n - 1, num=min(n, k)).astype(np.int64))


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


def grad_table(eng, G, strip=""):
    rows = {}
    for name, (off, shape) in eng.layout.entries.items():
        key = name[len(strip):] if strip and name.startswith(strip) else name
        g = eng.param_view("student", name, grad=True).reshape(-1).double().cpu()
        if "gnone/" + key in G:
            assert float(g.abs().max()) == 0.0, name            # grad None in the reference <-> exactly untouched here
            continue
        if "gsamp/" + key not in G:
            continue
        gn = float(G["gnorm/" + key])
        rows[name] = (rel(g[sample_idx(g.numel())].numpy(), G["gsamp/" + key]), float(g.norm()) / gn - 1.0, g.numel())
    return rows


def test_depth2_blocks_vs_reference_golden():
    G = load("clip_depth2_blocks")
    eng = AtstEngine("small", depth=2, drop_path_rate=0.0)
    eng.load_weights(O.recipe_weights("small", depth=2, seed=3))
    S = int(G["S"])
    mel = O.recipe_mel(S, 1001, seed=5).cuda()
    length = torch.from_numpy(G["length"])
    ep = eng._pass("student", S, 1001, True, 0)
    valid = eng._valid(length, 1)
    assert torch.equal(valid.cpu() - 1, torch.from_numpy(G["patch_length"]).int())           # integer: bit-exact
    out = ep.forward(mel, valid, None, None)
    tok = ep.tokens().reshape(S, 256, 384)[:, :251].cpu().numpy()
    assert rel(tok[:, ::10, ::4], G["tokens"]) < 5e-3
    assert float(ep.tokens().reshape(S, 256, 384)[:, 251:].abs().max()) == 0.0               # pad rows stay zero
    b0 = ep.block_out(0).reshape(S, 256, 384)[:, :251].cpu().numpy()
    b1 = ep.block_out(1).reshape(S, 256, 384)[:, :251].cpu().numpy()
    assert rel(b0[:, ::10, ::4], G["block0"]) < 1e-2
    assert rel(b1[:, ::10, ::4], G["block1"]) < 1e-2
    cls = out.float().reshape(S, 256, 384)[:, 0].cpu().numpy()
    assert rel(cls, G["cls"]) < 1e-2


def test_encoder_gradient_vs_reference_golden():
    """Smooth objective L = sum(CLS * R): full-depth encoder fwd + bwd, ragged lengths, injected DropPath."""
    G = load("clip_encoder_grad")
    S = int(G["S"])
    eng = AtstEngine("small")
    eng.load_weights(O.recipe_weights("small", seed=21))
    ep = eng._pass("student", S, 1001, True, 0)
    out = ep.forward(O.recipe_mel(S, 1001, seed=23).cuda(), eng._valid(torch.from_numpy(G["length"]), 1), None,
                     eng.drop_path_scales(S, torch.from_numpy(G["keep"])))
    cls = out.float().reshape(S, 256, 384)[:, 0].cpu().numpy()
    assert rel(cls, G["cls"]) < 1e-2
    R = torch.from_numpy(np.random.default_rng(29).standard_normal((S, 384)).astype(np.float32)).cuda()
    eng.g32.zero_(); ep.dout.zero_()
    rows = (torch.arange(S, dtype=torch.int32, device="cuda") * 256).contiguous()
    hip.call("atst_scatter_rows_bf16", hip.ptr(R), hip.ptr(rows), S, 384, hip.ptr(ep.dout), hip.stream())
    ep.backward()
    tab = {k: v for k, v in grad_table(eng, G, strip="encoder.").items() if k.startswith("encoder.")}
    assert len(tab) == 138                                             # every encoder tensor except mask_embed
    mean = sum(r * n for r, _, n in tab.values()) / sum(n for _, _, n in tab.values())
    worst = max(tab.items(), key=lambda kv: kv[1][0])
    print(f"\n[encoder grad] weighted mean rel-L2 {mean:.3e}; worst {worst[0]} {worst[1][0]:.3e}")
    assert mean < 1.0e-2 and worst[1][0] < 2.4e-2            # measured 6.9e-3 / 1.57e-2 (pos_embed) (x1.5)
    assert max(abs(nerr) for _, nerr, _ in tab.values()) < 1e-2


def test_encoder_gradient_vs_bf16_emulating_oracle():
    """The same smooth objective against the oracle's bf16-EMULATION mode (oracle.emulate_bf16 rounds exactly where the HIP path rounds:
    GEMM operands, saved activations, gradient operands).  What is left is accumulation order and the few places the emulation
    cannot mirror (fp32 softmax statistics vs the kernels' exp2 path, LayerNorm in the GEMM epilogues), so the bound is an order of
    magnitude below the 1e-2 of the fp32 comparison: it pins the production driver's own op order -- csrc/engine.hip, the fused
    residual + LayerNorm and LayerNorm-backward epilogues, attention forward / backward -- not just its wiring."""
    G = load("clip_encoder_grad")
    S = int(G["S"])
    W = O.recipe_weights("small", seed=21)
    eng = AtstEngine("small")
    eng.load_weights(W)
    ep = eng._pass("student", S, 1001, True, 0)
    mel, length, keep = O.recipe_mel(S, 1001, seed=23), torch.from_numpy(G["length"]), torch.from_numpy(G["keep"])
    out = ep.forward(mel.cuda(), eng._valid(length, 1), None, eng.drop_path_scales(S, keep))
    R = torch.from_numpy(np.random.default_rng(29).standard_normal((S, 384)).astype(np.float32))
    eng.g32.zero_(); ep.dout.zero_()
    rows = (torch.arange(S, dtype=torch.int32, device="cuda") * 256).contiguous()
    Rc = R.cuda()
    hip.call("atst_scatter_rows_bf16", hip.ptr(Rc), hip.ptr(rows), S, 384, hip.ptr(ep.dout), hip.stream())
    ep.backward()
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in W.items() if k.startswith("student.encoder.") and v.dtype == torch.float32}
    Wl = {k: (leaves[k] if k in leaves else v) for k, v in W.items()}
    with O.emulate_bf16():
        cls = O.encoder_forward(Wl, "student.encoder.", mel, length, "small", keep=keep, drop_path_rate=0.1)
        # the upstream gradient enters as a bf16 operand on the HIP side (ep.dout is bf16): same rounding here
        (cls * R.to(torch.bfloat16).float()).sum().backward()
    cls_h = out.float().reshape(S, 256, 384)[:, 0].cpu()
    fwd = rel(cls_h.numpy(), cls.detach().numpy())
    num = den = 0.0
    worst = ("", 0.0)
    for k, v in leaves.items():
        if v.grad is None:
            continue
        name = k[len("student."):]
        g = eng.param_view("student", name, grad=True).detach().cpu()
        r = rel(g.reshape(-1).numpy(), v.grad.reshape(-1).numpy())
        num += r * v.numel(); den += v.numel()
        if r > worst[1]:
            worst = (name, r)
    print(f"\n[encoder grad vs bf16-emulating oracle] CLS rel-L2 {fwd:.3e}; weighted mean rel-L2 {num / den:.3e}; worst {worst[0]} {worst[1]:.3e}")
    # measured 4.1e-3 / 5.0e-3 / 6.6e-3 (pos_embed) (x1.5).  Against the fp32 golden the same run gives 6.9e-3 mean / 1.6e-2 worst: the
    # emulation removes the systematic part (where the roundings sit), what remains is that two bf16 realisations of a 12-layer
    # network decorrelate -- each rounding decision depends on the last bits of the accumulation order.
    assert fwd < 6.2e-3 and num / den < 7.5e-3 and worst[1] < 1.0e-2, (fwd, num / den, worst)


def _emulation_twin(arch, depth, frame, seed_w, seed_x, seed_r, S=6):
    """HIP encoder gradient of a smooth objective against the oracle's bf16-EMULATION mode on the same weights / inputs / DropPath decisions:
    clip encoders: L = sum(CLS * R) ; frame encoders: L = sum(LN(x)[masked & valid rows] * R) with a block mask per sequence (mask tokens substituted)."""
    d = 768 if arch == "base" else 384
    W = O.recipe_weights(arch, depth=depth, frame=frame, seed=seed_w)
    eng = AtstEngine(arch, depth=depth, frame=frame)
    eng.load_weights(W)
    ep = eng._pass("student", S, 1001, True, 0)
    mel = O.recipe_mel(S, 1001, seed=seed_x)
    length = torch.tensor([1001, 1001, 702, 941, 523, 1001][:S])
    g = torch.Generator().manual_seed(seed_r)
    rates = [float(v) for v in torch.linspace(0, 0.1, depth)]
    keep = torch.ones(depth, 2, S)
    for i in range(1, depth):
        keep[i] = torch.floor((1.0 - rates[i]) + torch.rand(2, S, generator=g))
    use_cls = 0 if frame else 1
    valid = eng._valid(length, use_cls, ep.n_tok + use_cls)
    mk = rowflag = None
    if frame:
        rs = np.random.RandomState(seed_r)
        mk = torch.from_numpy(np.stack([O.block_mask(250, 0.65, 5, rng=rs) for _ in range(S)]))
        rows, rowflag = eng._frame_rows(mk.bool(), valid, ep.RS, True)
    else:
        rows = (torch.arange(S, dtype=torch.int32, device="cuda") * ep.RS).contiguous()
    out = ep.forward(mel.cuda(), valid, rowflag, eng.drop_path_scales(S, keep))
    nrow = int(rows.numel())
    R = torch.from_numpy(np.random.default_rng(seed_r).standard_normal((nrow, d)).astype(np.float32))
    eng.g32.zero_(); ep.dout.zero_()
    Rc = R.cuda()
    hip.call("atst_scatter_rows_bf16", hip.ptr(Rc), hip.ptr(rows), nrow, d, hip.ptr(ep.dout), hip.stream())
    ep.backward()
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in W.items() if k.startswith("student.encoder.") and v.dtype == torch.float32}
    Wl = {k: (leaves[k] if k in leaves else v) for k, v in W.items()}
    with O.emulate_bf16():
        y = O.encoder_forward(Wl, "student.encoder.", mel, length, arch, depth=depth, use_cls=not frame, mask_index=mk, mask_input=True, keep=keep, drop_path_rate=0.1)
        (y * R.to(torch.bfloat16).float()).sum().backward()                # the upstream gradient enters as a bf16 operand on the HIP side
    y_h = out.float()[rows.long()].cpu()
    assert y_h.shape == y.shape
    fwd = rel(y_h.numpy(), y.detach().numpy())
    num = den = 0.0
    worst = ("", 0.0)
    for k, v in leaves.items():
        if v.grad is None or float(v.grad.norm()) == 0.0:
            continue
        name = k[len("student."):]
        gh = eng.param_view("student", name, grad=True).detach().cpu()
        r = rel(gh.reshape(-1).numpy(), v.grad.reshape(-1).numpy())
        num += r * v.numel(); den += v.numel()
        if r > worst[1]:
            worst = (name, r)
    return fwd, num / den, worst


@pytest.mark.parametrize("arch,depth,frame,bounds", [("base", 3, False, (5.9e-3, 6.7e-3, 9.1e-3)), ("small", 12, True, (6.4e-3, 7.4e-3, 9.6e-3)), ("base", 2, True, (5.7e-3, 6.5e-3, 8.1e-3))])   # measured 3.9e-3 / 4.5e-3 / 6.1e-3 ; 4.3e-3 / 4.9e-3 / 6.4e-3 ; 3.8e-3 / 4.3e-3 / 5.4e-3 (x1.5)
def test_encoder_gradient_emulation_twins(arch, depth, frame, bounds):
    """the bf16-emulating-oracle comparison existed only for the d = 384 clip encoder.  Here its twins: the d = 768 clip encoder, the
    ATST-Frame encoder (mask-token substitution, ragged masked-row gather, norm_frame) at d = 384 (12 layers) and d = 768 -- the HIP gradient against an
    oracle that rounds where HIP rounds, so that a real error in one small tensor cannot hide inside the bf16-vs-fp32 tolerance of the golden tests."""
    fwd, mean, worst = _emulation_twin(arch, depth, frame, 101, 103, 107)
    print(f"\n[emulation twin {arch} depth {depth} {'frame' if frame else 'clip'}] output rel-L2 {fwd:.3e}; gradient weighted mean rel-L2 {mean:.3e}; worst {worst[0]} {worst[1]:.3e}")
    assert fwd < bounds[0] and mean < bounds[1] and worst[1] < bounds[2], (fwd, mean, worst)


def test_base_arch_encoder_vs_oracle():
    """ATST-base geometry (d = 768, 12 heads; audio_transformer.py:372-374), depth 3, ragged lengths, injected DropPath:
    forward CLS and the gradient of the smooth objective sum(CLS * R) against the CPU oracle's autograd.  Exercises the
    C = 768 code paths (separate LayerNorm kernels, 768 / 2304 / 3072-wide GEMM tiles, 12-head attention loop)."""
    S, depth = 6, 3
    W = O.recipe_weights("base", depth=depth, seed=31)
    eng = AtstEngine("base", depth=depth)
    eng.load_weights(W)
    mel = O.recipe_mel(S, 1001, seed=33)
    length = torch.tensor([1001, 1001, 777, 640, 1001, 405])
    keep = (torch.rand(depth, 2, S, generator=torch.Generator().manual_seed(7)) > 0.3).float()
    keep[0] = 1.0                                                       # block 0 has no DropPath (rate 0)
    leaves = {k: v.requires_grad_(True) for k, v in W.items() if k.startswith("student.encoder.") and v.dtype == torch.float32}
    cls_o = O.encoder_forward(W, "student.encoder.", mel, length, "base", depth, keep=keep)
    R = torch.from_numpy(np.random.default_rng(35).standard_normal((S, 768)).astype(np.float32))
    (cls_o * R).sum().backward()
    ep = eng._pass("student", S, 1001, True, 0)
    out = ep.forward(mel.cuda(), eng._valid(length, 1), None, eng.drop_path_scales(S, keep))
    cls = out.float().reshape(S, 256, 768)[:, 0].cpu()
    assert rel(cls.numpy(), cls_o.detach().numpy()) < 1e-2
    eng.g32.zero_(); ep.dout.zero_()
    rows = (torch.arange(S, dtype=torch.int32, device="cuda") * 256).contiguous()
    hip.call("atst_scatter_rows_bf16", hip.ptr(R.cuda()), hip.ptr(rows), S, 768, hip.ptr(ep.dout), hip.stream())
    ep.backward()
    num = den = 0.0
    worst = ("", 0.0)
    for name in eng.layout.entries:
        if not name.startswith("encoder.") or name == "encoder.mask_embed":
            continue
        g = eng.param_view("student", name, grad=True).double().cpu()
        go = leaves["student." + name].grad.double()
        r = float((g - go).norm() / (go.norm() + 1e-30))
        num += r * g.numel(); den += g.numel()
        if r > worst[1]:
            worst = (name, r)
    print(f"\n[base encoder grad] weighted mean rel-L2 {num / den:.3e}; worst {worst[0]} {worst[1]:.3e}")
    assert num / den < 1.0e-2 and worst[1] < 2.4e-2             # measured 6.9e-3 / 1.6e-2 (x1.5)


def test_base_encoder_gradient_vs_reference_golden():
    """ATST-base geometry (d = 768, 12 heads: what AST_base builds, audio_transformer.py:371-374) pinned to the IMPORTED REFERENCE
    (tests/golden/base_depth3_encoder_grad.npz): block activations, CLS and the gradient of sum(CLS * R) at depth 3,
    ragged lengths, the reference's recorded DropPath draws.  Same bounds as the d = 384 golden test."""
    G = load("base_depth3_encoder_grad")
    S, depth = int(G["S"]), int(G["depth"])
    eng = AtstEngine("base", depth=depth)
    eng.load_weights(O.recipe_weights("base", depth=depth, seed=31))
    length = torch.from_numpy(G["length"])
    ep = eng._pass("student", S, 1001, True, 0)
    valid = eng._valid(length, 1)
    assert torch.equal(valid.cpu() - 1, torch.from_numpy(G["patch_length"]).int())           # integer: bit-exact
    out = ep.forward(O.recipe_mel(S, 1001, seed=33).cuda(), valid, None, eng.drop_path_scales(S, torch.from_numpy(G["keep"])))
    for i in range(depth):
        b = ep.block_out(i).reshape(S, 256, 768)[:, :251].cpu().numpy()
        assert rel(b[:, ::10, ::8], G[f"block{i}"]) < 1e-2, i
    cls = out.float().reshape(S, 256, 768)[:, 0].cpu().numpy()
    assert rel(cls, G["cls"]) < 1e-2
    R = torch.from_numpy(np.random.default_rng(35).standard_normal((S, 768)).astype(np.float32)).cuda()
    eng.g32.zero_(); ep.dout.zero_()
    rows = (torch.arange(S, dtype=torch.int32, device="cuda") * 256).contiguous()
    hip.call("atst_scatter_rows_bf16", hip.ptr(R), hip.ptr(rows), S, 768, hip.ptr(ep.dout), hip.stream())
    ep.backward()
    tab = {k: v for k, v in grad_table(eng, G, strip="encoder.").items() if k.startswith("encoder.")}
    assert len(tab) == 6 + 11 * depth                                  # every encoder tensor except mask_embed
    mean = sum(r * n for r, _, n in tab.values()) / sum(n for _, _, n in tab.values())
    worst = max(tab.items(), key=lambda kv: kv[1][0])
    print(f"\n[base encoder grad vs reference golden] weighted mean rel-L2 {mean:.3e}; worst {worst[0]} {worst[1][0]:.3e}")
    assert mean < 1.0e-2 and worst[1][0] < 2.4e-2
    assert max(abs(nerr) for _, nerr, _ in tab.values()) < 1e-2


def test_base_fp8_encoder_gradient_vs_reference_golden():
    """BASELINE.json configs[4] against the IMPORTED REFERENCE itself (not the emulating oracle): the same d = 768 golden as above, computed by the
    fp8 engine -- step 1 with the e4m3 forward and the recording bf16 backward, step 2 with all 12 GEMMs of every block on e4m3 operands (forward, dgrads,
    weight gradients; delayed scales from step 1).  The reference is fp32: what is measured here is the whole e4m3 staircase (3 mantissa bits, 2^-4
    relative per operand), so the bounds are ~10x the bf16 ones -- 1.5x what was measured, stated below."""
    G = load("base_depth3_encoder_grad")
    S, depth = int(G["S"]), int(G["depth"])
    eng = AtstEngine("base", depth=depth, fp8=True)
    eng.load_weights(O.recipe_weights("base", depth=depth, seed=31))
    length = torch.from_numpy(G["length"])
    ep = eng._pass("student", S, 1001, True, 0)
    valid = eng._valid(length, 1)
    R = torch.from_numpy(np.random.default_rng(35).standard_normal((S, 768)).astype(np.float32)).cuda()
    rows = (torch.arange(S, dtype=torch.int32, device="cuda") * 256).contiguous()
    res = []
    for step in range(2):
        assert eng.fp8_bwd_state == step + 1 and eng.fp8_wgrad_mode() == (3, 2)[step]
        out = ep.forward(O.recipe_mel(S, 1001, seed=33).cuda(), valid, None, eng.drop_path_scales(S, torch.from_numpy(G["keep"])))
        assert ep.e.fp8_lean == (0, 2)[step]                           # step 2 does not write the bf16 LayerNorm / GELU copies
        cls = out.float().reshape(S, 256, 768)[:, 0].cpu().numpy()
        eng._fp8_after_forward()
        eng.g32.zero_(); ep.dout.zero_()
        hip.call("atst_scatter_rows_bf16", hip.ptr(R), hip.ptr(rows), S, 768, hip.ptr(ep.dout), hip.stream())
        ep.backward()
        eng._fp8_after_backward()
        tab = {k: v for k, v in grad_table(eng, G, strip="encoder.").items() if k.startswith("encoder.")}
        mean = sum(r * n for r, _, n in tab.values()) / sum(n for _, _, n in tab.values())
        worst = max(tab.items(), key=lambda kv: kv[1][0])
        res.append((rel(cls, G["cls"]), mean, worst))
        print(f"\n[base fp8 encoder grad vs reference golden, step {step + 1}: {('e4m3 forward + bf16 backward', 'all 12 GEMMs e4m3')[step]}] CLS rel-L2 {res[-1][0]:.3e}; "
              f"gradient weighted mean rel-L2 {mean:.3e}; worst {worst[0]} {worst[1][0]:.3e}")
    assert res[0][0] < FP8_CLS and res[1][0] < FP8_CLS
    assert res[0][1] < FP8_GRAD_FWD and res[1][1] < FP8_GRAD_ALL and res[1][2][1][0] < FP8_GRAD_WORST


@pytest.mark.parametrize("fused", [True, False])
def test_small_fp8_encoder_gradient_vs_reference_golden(fused):
    """(fused: the LayerNorms of the e4m3 step inside the GEMM epilogues -- the default since round 6 -- or as separate passes, tuning hook 2110.)
    All-e4m3 + fp8_lean at d = 384 (round 6): the full-depth ATST-small encoder golden of the IMPORTED REFERENCE (clip_encoder_grad.npz:
    12 layers, ragged lengths, recorded DropPath draws) computed by the fp8 engine -- step 1 e4m3 forward + recording bf16 backward (unfused LayerNorm
    backward: its kernel records the amax of the gradient operands), step 2 all 12 GEMMs of every block on e4m3 operands: dgrads, the NP = 256 attention
    backward writing dqkv as e4m3 only, weight gradients through gemm_tn8's half-valid 256 x 256 edge tiles (384 / 1152 / 1536 are multiples of 128, not
    256), no bf16 LayerNorm / GELU copies.  Bounds = measured x 1.5 (the e4m3 staircase over 12 layers against an fp32 reference)."""
    G = load("clip_encoder_grad")
    S = int(G["S"])
    hip.load().atst_tune_gemm_variant(2111 if fused else 2110)
    try:
        res, eng = _small_fp8_two_steps(G, S)
    finally:
        hip.load().atst_tune_gemm_variant(2111)
    assert sum(eng.fp8_saturation().values()) == 0
    assert res[0][0] < FP8S_CLS and res[1][0] < FP8S_CLS
    assert res[0][1] < FP8S_GRAD_FWD and res[1][1] < FP8S_GRAD_ALL and res[1][2][1][0] < FP8S_GRAD_WORST


def _small_fp8_two_steps(G, S, keep_grads=None):
    eng = AtstEngine("small", fp8=True)
    eng.load_weights(O.recipe_weights("small", seed=21))
    ep = eng._pass("student", S, 1001, True, 0)
    valid = eng._valid(torch.from_numpy(G["length"]), 1)
    R = torch.from_numpy(np.random.default_rng(29).standard_normal((S, 384)).astype(np.float32)).cuda()
    rows = (torch.arange(S, dtype=torch.int32, device="cuda") * 256).contiguous()
    res = []
    for step in range(2):
        assert eng.fp8_bwd_state == step + 1 and eng.fp8_wgrad_mode() == (3, 2)[step]
        out = ep.forward(O.recipe_mel(S, 1001, seed=23).cuda(), valid, None, eng.drop_path_scales(S, torch.from_numpy(G["keep"])))
        assert ep.e.fp8_lean == (0, 2)[step]
        cls = out.float().reshape(S, 256, 384)[:, 0].cpu().numpy()
        eng._fp8_after_forward()
        eng.g32.zero_(); ep.dout.zero_()
        hip.call("atst_scatter_rows_bf16", hip.ptr(R), hip.ptr(rows), S, 384, hip.ptr(ep.dout), hip.stream())
        ep.backward()
        eng._fp8_after_backward()
        tab = {k: v for k, v in grad_table(eng, G, strip="encoder.").items() if k.startswith("encoder.")}
        assert len(tab) == 138
        mean = sum(r * n for r, _, n in tab.values()) / sum(n for _, _, n in tab.values())
        worst = max(tab.items(), key=lambda kv: kv[1][0])
        res.append((rel(cls, G["cls"]), mean, worst))
        print(f"\n[small fp8 encoder grad vs reference golden, step {step + 1}: {('e4m3 forward + bf16 backward', 'all 12 GEMMs e4m3')[step]}] CLS rel-L2 {res[-1][0]:.3e}; "
              f"gradient weighted mean rel-L2 {mean:.3e}; worst {worst[0]} {worst[1][0]:.3e}")
        if keep_grads is not None:
            keep_grads.append((torch.from_numpy(cls.copy()), eng.g32.detach().clone()))
    return res, eng


def test_small_fp8_fused_layernorm_epilogues_vs_separate_passes():
    """The same two steps with the LayerNorms inside the GEMM epilogues (default) and as separate passes (hook 2110), against EACH OTHER: the forward differs
    only where a LayerNorm output sits on an e4m3 rounding boundary (the two kernels sum the row statistics in different orders); the e4m3 backward also
    because the fused epilogue takes the dgrad accumulators in fp32 where the separate pass reads them rounded to bf16 -- a bf16-level change that moves
    e4m3 codes of the gradient operands.  Bounds = measured x 1.5, an order of magnitude inside the distance of either from the fp32 reference."""
    G = load("clip_encoder_grad")
    S = int(G["S"])
    runs = {}
    for fused in (True, False):
        hip.load().atst_tune_gemm_variant(2111 if fused else 2110)
        try:
            kept = []
            _small_fp8_two_steps(G, S, keep_grads=kept)
            runs[fused] = kept
        finally:
            hip.load().atst_tune_gemm_variant(2111)
    out = []
    for step in range(2):
        (ca, ga), (cb, gb) = runs[True][step], runs[False][step]
        out.append((rel(ca.numpy(), cb.numpy()), float((ga - gb).norm() / gb.norm())))
    print(f"\n[small fp8: fused LayerNorm epilogues vs separate passes] step 1 (e4m3 forward, bf16 recording backward -- both unfused in the backward) CLS {out[0][0]:.3e} grad {out[0][1]:.3e}; "
          f"step 2 (all e4m3) CLS {out[1][0]:.3e} grad {out[1][1]:.3e}")
    assert out[0][0] < FUSED8_CLS and out[1][0] < FUSED8_CLS and out[0][1] < FUSED8_GRAD1 and out[1][1] < FUSED8_GRAD2


FUSED8_CLS, FUSED8_GRAD1, FUSED8_GRAD2 = 3e-2, 7.6e-3, 8.4e-2     # measured 8.9e-3 (step 1) / 1.98e-2 (step 2: its scales come from step 1's amax) ; 5.1e-3 ; 5.6e-2 -- x1.5


FP8S_CLS, FP8S_GRAD_FWD, FP8S_GRAD_ALL, FP8S_GRAD_WORST = 0.13, 0.12, 0.15, 0.20   # measured 8.8e-2 ; 8.0e-2 ; 9.9e-2 ; 0.132 (blocks.11.norm2.weight) -- x1.5 (12 layers; the d = 768 golden has 3).  (Before the du-amax fix of the 128 x 128 dGELU kernel: 0.137 / 0.216.)


FP8_CLS, FP8_GRAD_FWD, FP8_GRAD_ALL, FP8_GRAD_WORST = 0.12, 0.11, 0.14, 0.26   # measured 7.9e-2 ; 7.2e-2 ; 9.1e-2 ; 0.169 (pos_embed) -- x1.5


def test_base_step_vs_reference_golden():
    """One 2-view training step of an ATST("base")-shaped model at depth 2 against the imported reference (MultiCropWrapper + ByolLoss
    around AST(768, 12 heads), tests/golden/base_2views_depth2.npz): loss, head outputs, gradients, BN buffers, EMA."""
    G = load("base_2views_depth2")
    B, depth = int(G["B"]), int(G["depth"])
    eng = AtstEngine("base", depth=depth, drop_path_rate=0.1)
    eng.load_weights(O.recipe_weights("base", depth=depth, seed=41))
    mels = [O.recipe_mel(B, int(w), seed=43 + i) for i, w in enumerate(G["widths"])]
    lens = [torch.from_numpy(l) for l in G["lengths"]]
    loss, std_s, std_t = eng.forward(mels, lens, None, [torch.from_numpy(G["keep_t0"])], [torch.from_numpy(G["keep_s0"])])
    eng.backward()
    s_out, t_out = eng.last_outputs
    rs, rt = rel(s_out.cpu().numpy()[:8], G["student_out"]), rel(t_out.cpu().numpy()[:8], G["teacher_out"])
    print(f"\n[base step] loss {loss.item():.6f} (ref {float(G['loss']):.6f}) std_s {std_s.item():.5f}/{float(G['std_s']):.5f} out rel {rs:.2e} {rt:.2e}")
    assert abs(loss.item() - float(G["loss"])) < 5e-3
    assert abs(std_s.item() - float(G["std_s"])) < 2e-3 and abs(std_t.item() - float(G["std_t"])) < 2e-3
    assert rs < 2.5e-2 and rt < 2.5e-2
    tab = grad_table(eng, G)
    assert len(tab) == 6 + 11 * depth + 8                              # encoder (minus mask_embed, grad None) + two heads
    assert tab["predictor.3.weight"][0] < 3e-2 and tab["predictor.1.weight"][0] < 3e-2
    for k, (r, nerr, n) in tab.items():
        if k in CANCELLING:
            continue
        assert abs(nerr) < 5e-2, (k, nerr)
        assert r < 0.19, (k, r)                                        # ReLU-gate flips (see the module docstring); B = 16
    for k in ("student.projector.1.running_mean", "student.projector.1.running_var", "teacher.projector.1.running_var"):
        net, which, _, buf = k.split(".")
        got = eng.bn_buffers[f"{net}.{which}"][buf].cpu()[sample_idx(4096)].numpy()
        assert rel(got, G["bn/" + k]) < 2e-2, k
    eng.ema_update(0.99)
    for k in ("teacher.encoder.pos_embed", "teacher.encoder.blocks.1.mlp.fc1.weight", "teacher.projector.0.weight"):
        v = eng.param_view("teacher", k[len("teacher."):]).reshape(-1).cpu()
        assert rel(v[sample_idx(v.numel())].numpy(), G["ema/" + k]) < 1e-6, k


@pytest.mark.parametrize("S", [5, 64])                  # 64 sequences: the packed layout (row stride = token count < tile rows)
@pytest.mark.parametrize("width,NP", [(401, 128), (201, 64), (101, 32), (41, 32)])
def test_short_clip_geometries_vs_oracle(width, NP, S):
    """4 s / 2 s / 1 s / 0.4 s inputs (token tiles of 128 / 64 / 32 rows; the generic attention kernels), depth 2, ragged
    lengths: CLS and the gradient of sum(CLS * R) against the CPU oracle's autograd.  With S = 64 the engine packs the
    sequences (EncoderPass.RS = tokens per sequence instead of the tile-aligned NP)."""
    depth = 2
    W = O.recipe_weights("small", depth=depth, seed=41)
    eng = AtstEngine("small", depth=depth, drop_path_rate=0.0)
    eng.load_weights(W)
    mel = O.recipe_mel(S, width, seed=43)
    length = torch.tensor([width, width - 4, max(width // 2, 8), width - 1, max(width // 3, 5)] * ((S + 4) // 5))[:S]
    leaves = {k: v.requires_grad_(True) for k, v in W.items() if k.startswith("student.encoder.") and v.dtype == torch.float32}
    cls_o = O.encoder_forward(W, "student.encoder.", mel, length, "small", depth, drop_path_rate=0.0)
    R = torch.from_numpy(np.random.default_rng(45).standard_normal((S, 384)).astype(np.float32))
    (cls_o * R).sum().backward()
    ep = eng._pass("student", S, width, True, 0)
    assert ep.NP == NP and ep.RS == (ep.n_tok + 1 if S == 64 else NP)
    out = ep.forward(mel.cuda(), eng._valid(length, 1), None, None)
    cls = out.float().reshape(S, ep.RS, 384)[:, 0].cpu()
    assert rel(cls.numpy(), cls_o.detach().numpy()) < 1e-2
    eng.g32.zero_(); ep.dout.zero_()
    rows = (torch.arange(S, dtype=torch.int32, device="cuda") * ep.RS).contiguous()
    hip.call("atst_scatter_rows_bf16", hip.ptr(R.cuda()), hip.ptr(rows), S, 384, hip.ptr(ep.dout), hip.stream())
    ep.backward()
    num = den = 0.0
    worst = ("", 0.0)
    for name in eng.layout.entries:
        if not name.startswith("encoder.") or name == "encoder.mask_embed":
            continue
        g = eng.param_view("student", name, grad=True).double().cpu()
        go = leaves["student." + name].grad.double()
        r = float((g - go).norm() / (go.norm() + 1e-30))
        num += r * g.numel(); den += g.numel()
        if r > worst[1]:
            worst = (name, r)
    assert num / den < 1.5e-2 and worst[1] < 3e-2, (num / den, worst)


def test_local_views_fp8_e4m3_attention_outputs_vs_oracle():
    """Round 6: the 1 s local views take part in the e4m3 step -- the NP = 32 attention kernels write the e4m3 copy of their output (forward) and dqkv as e4m3
    only (fused backward), so the qkv dgrad and weight gradient of a PACKED pass (64 sequences x 26 rows) read e4m3 like those of the 10 s views and the pass
    runs `fp8_lean` = 2.  Depth 2, ragged lengths, CLS and the gradient of sum(CLS * R): step 1 (e4m3 forward, recording bf16 backward), step 2 (all e4m3)
    against the CPU oracle's fp32 autograd, and step 2 against the same step with the NP = 32 e4m3 outputs off (hook 412: quantisation pass + bf16 dqkv)."""
    depth, S, width = 2, 64, 101
    W = O.recipe_weights("small", depth=depth, seed=41)
    mel = O.recipe_mel(S, width, seed=43)
    length = torch.tensor([width, width - 4, max(width // 2, 8), width - 1, max(width // 3, 5)] * ((S + 4) // 5))[:S]
    leaves = {k: v.requires_grad_(True) for k, v in W.items() if k.startswith("student.encoder.") and v.dtype == torch.float32}
    cls_o = O.encoder_forward(W, "student.encoder.", mel, length, "small", depth, drop_path_rate=0.0)
    R = torch.from_numpy(np.random.default_rng(45).standard_normal((S, 384)).astype(np.float32))
    (cls_o * R).sum().backward()
    runs = {}
    for hook in (413, 412):
        hip.load().atst_tune_gemm_variant(hook)
        try:
            eng = AtstEngine("small", depth=depth, drop_path_rate=0.0, fp8=True)
            eng.load_weights(W)
            ep = eng._pass("student", S, width, True, 0)
            assert ep.NP == 32 and ep.RS == 26 and ep.M % 64 == 0
            rows = (torch.arange(S, dtype=torch.int32, device="cuda") * ep.RS).contiguous()
            res = []
            for step in range(2):
                out = ep.forward(mel.cuda(), eng._valid(length, 1), None, None)
                assert ep.e.fp8_lean == ((0, 2) if hook == 413 else (0, 1))[step]
                cls = out.float().reshape(S, ep.RS, 384)[:, 0].cpu()
                eng._fp8_after_forward()
                eng.g32.zero_(); ep.dout.zero_()
                hip.call("atst_scatter_rows_bf16", hip.ptr(R.cuda()), hip.ptr(rows), S, 384, hip.ptr(ep.dout), hip.stream())
                ep.backward()
                eng._fp8_after_backward()
                num = den = 0.0
                worst = ("", 0.0)
                for name in eng.layout.entries:
                    if not name.startswith("encoder.") or name == "encoder.mask_embed":
                        continue
                    g = eng.param_view("student", name, grad=True).double().cpu()
                    go = leaves["student." + name].grad.double()
                    r = float((g - go).norm() / (go.norm() + 1e-30))
                    num += r * g.numel(); den += g.numel()
                    if r > worst[1]:
                        worst = (name, r)
                res.append((rel(cls.numpy(), cls_o.detach().numpy()), num / den, worst, eng.g32.detach().clone()))
            assert sum(eng.fp8_saturation().values()) == 0
            runs[hook] = res
        finally:
            hip.load().atst_tune_gemm_variant(413)
    for hook, res in runs.items():
        for step, (c, m, w, _) in enumerate(res):
            print(f"\n[fp8 local views, NP = 32 e4m3 outputs {'on' if hook == 413 else 'off'}, step {step + 1}] CLS {c:.3e}; gradient mean {m:.3e} worst {w[0]} {w[1]:.3e}")
            assert c < LOC8_CLS and m < LOC8_MEAN and w[1] < LOC8_WORST, (hook, step, c, m, w)
    d1 = float((runs[413][0][3] - runs[412][0][3]).norm() / runs[412][0][3].norm())
    d2 = float((runs[413][1][3] - runs[412][1][3]).norm() / runs[412][1][3].norm())
    print(f"[fp8 local views] NP = 32 e4m3 outputs on vs off: step 1 {d1:.3e} (the same bf16 values quantised by the kernel or by a pass: identical), step 2 {d2:.3e}")
    assert d1 < 1e-5 and d2 < LOC8_ONOFF                 # (step 1: fp32 atomics order only)


LOC8_CLS, LOC8_MEAN, LOC8_WORST, LOC8_ONOFF = 0.107, 0.119, 0.137, 4.4e-2     # measured: CLS 7.1e-2 ; gradients 5.6e-2 (step 1) / 7.9e-2 (step 2) mean, 9.1e-2 worst (blocks.1.norm2.weight) ; on vs off 2.9e-2 (e4m3 qkv dgrad + weight gradient instead of bf16) -- x1.5


def test_split_backward_equals_single_call():
    """atst_encoder_bwd_part(0, s) + (1, s) -- the form backward() uses to start the gradient all-reduce of the upper blocks
    early -- produces the gradients of the single-call backward (up to fp32 atomic order)."""
    S, depth = 6, 4
    eng = AtstEngine("small", depth=depth, drop_path_rate=0.0)
    eng.load_weights(O.recipe_weights("small", depth=depth, seed=51))
    ep = eng._pass("student", S, 1001, True, 0)
    ep.forward(O.recipe_mel(S, 1001, seed=53).cuda(), eng._valid(torch.tensor([1001, 900, 1001, 640, 1001, 333]), 1), None, None)
    R = torch.from_numpy(np.random.default_rng(55).standard_normal((S, 384)).astype(np.float32)).cuda()
    rows = (torch.arange(S, dtype=torch.int32, device="cuda") * 256).contiguous()
    grads = []
    for split in (None, 1, 2, 3, 0, 4):
        eng.g32.zero_(); ep.dout.zero_()
        hip.call("atst_scatter_rows_bf16", hip.ptr(R), hip.ptr(rows), S, 384, hip.ptr(ep.dout), hip.stream())
        if split is None:
            ep.backward()
        else:
            ep.backward_part(0, split); ep.backward_part(1, split)
        grads.append(eng.g32.clone())
    for g in grads[1:]:
        assert float((g - grads[0]).norm() / grads[0].norm()) < 1e-6


def test_head_and_loss_backward_given_same_features():
    """Projector + predictor + loss forward/backward on fixed features vs the oracle's autograd on the same features."""
    B, Cdim = 48, 384
    W = O.recipe_weights("small", depth=1, seed=9)
    eng = AtstEngine("small", depth=1)
    eng.load_weights(W)
    g = torch.Generator().manual_seed(3)
    base = torch.randn(1, Cdim, generator=g)
    fs = (base + 0.7 * torch.randn(2 * B, Cdim, generator=g)).contiguous()         # student CLS rows (2 views)
    ft = (base + 0.7 * torch.randn(2 * B, Cdim, generator=g)).contiguous()
    t_out = eng.heads["teacher.projector"].forward(ft.cuda(), False)
    z = eng.heads["student.projector"].forward(fs.cuda(), True)
    s_out = eng.heads["student.predictor"].forward(z, True)
    acc, ds, stats = torch.empty(1, device="cuda"), torch.empty_like(s_out), torch.empty(4, 256, device="cuda")
    hip.call("atst_byol_loss_f32", hip.ptr(s_out), hip.ptr(t_out), B, 2, 256, hip.ptr(acc), hip.ptr(ds), hip.ptr(stats), hip.stream())
    eng.g32.zero_()
    dfeat = eng.heads["student.projector"].backward(eng.heads["student.predictor"].backward(ds))
    loss_h = 2.0 - 2.0 * acc.item() / (2 * B)
    # oracle
    leaves = {k: v.requires_grad_(True) for k, v in W.items() if k.startswith(("student.projector", "student.predictor"))
              and v.dtype == torch.float32 and "running" not in k}
    fso = fs.clone().requires_grad_(True)
    with torch.no_grad():
        t_o = O.mlp_head(W, "teacher.projector.", ft)
    s_o = O.mlp_head(W, "student.predictor.", O.mlp_head(W, "student.projector.", fso))
    loss_o, _, _ = O.byol_loss(s_o, t_o, 2)
    loss_o.backward()
    assert abs(loss_h - loss_o.item()) < 3e-4
    assert rel(s_out.cpu().numpy(), s_o.detach().numpy()) < 5e-3 and rel(t_out.cpu().numpy(), t_o.numpy()) < 5e-3
    assert rel(dfeat.cpu().numpy(), fso.grad.numpy()) < 1.5e-2
    for k, v in leaves.items():
        got = eng.param_view("student", k[len("student."):], grad=True).cpu().numpy()
        assert rel(got, v.grad.numpy()) < 1.5e-2, k


def test_two_stream_step_equals_one_stream_step():
    """The local-view groups run beside the teacher pass (forward) and beside the global-view group's backward on a second HIP stream
    (AtstEngine.overlap_local_teacher, default on).  Same weights, same inputs, same DropPath decisions, stream on / off: the forward is made of
    the same launches in a different interleaving -- loss, head outputs and BatchNorm buffers must be bit-identical --, the gradients differ only
    by the order of their fp32 atomics (measured 2e-7 of the gradient norm).  6 crops, B = 8, three steps (optimizer + EMA in between)."""
    B, widths = 8, [1001, 1001, 101, 101, 101, 101]
    W = O.recipe_weights("small", depth=3, seed=21)
    gen = torch.Generator().manual_seed(5)
    keep_t = [(torch.rand(3, 2, 2 * B, generator=gen) > 0.1).float()]
    keep_s = [(torch.rand(3, 2, 2 * B, generator=gen) > 0.1).float(), (torch.rand(3, 2, 4 * B, generator=gen) > 0.1).float()]
    for k in keep_t + keep_s:
        k[0] = 1.0                                              # block 0 never drops (rate 0)
    runs = {}
    for on in (False, True):
        eng = AtstEngine("small", depth=3, ncrops=6, drop_path_rate=0.1)
        eng.overlap_local_teacher = on
        eng.load_weights(W)
        rec = []
        for step in range(3):
            mels = [O.recipe_mel(B, w, seed=100 * step + i).cuda() for i, w in enumerate(widths)]
            lens = [torch.full((B,), w) for w in widths]
            loss, _, _ = eng.forward(mels, lens, None, keep_t, keep_s)
            eng.backward()
            rec.append((float(loss), eng.last_outputs[0].clone(), eng.last_outputs[1].clone(), eng.g32.clone()))
            if step == 0:
                bn_first = {k: {b: t.clone() for b, t in v.items()} for k, v in eng.bn_buffers.items()}
            eng.optimizer_step(1e-3, 0.04, 0.99)
        torch.cuda.synchronize()
        runs[on] = (rec, eng.p32.clone(), eng.t32.clone(), bn_first)
    (r0, p0, t0, bn0), (r1, p1, t1, bn1) = runs[False], runs[True]
    l0, s0, to0, g0 = r0[0]
    l1, s1, to1, g1 = r1[0]
    assert l0 == l1 and torch.equal(s0, s1) and torch.equal(to0, to1)              # step 1: identical weights -> identical forward, bit for bit
    e = float((g0 - g1).norm() / g0.norm())
    print(f"\n[two streams vs one] step-1 loss {l0:.6f} == {l1:.6f}; gradient rel diff {e:.2e}; losses {[r[0] for r in r0]} / {[r[0] for r in r1]}")
    assert e < 2e-5
    for k in bn0:                                                                    # BatchNorm buffers after the first forward: same launches, same bits
        for b in bn0[k]:
            assert torch.equal(bn0[k][b], bn1[k][b]), (k, b)
    for (la, *_), (lb, *_) in zip(r0[1:], r1[1:]):
        assert abs(la - lb) < 2e-3                                                   # later steps: Adam turns the 1e-7 gradient noise into O(lr) parameter noise
    assert float((p0 - p1).abs().max()) < 2.5e-3 * 3 and float((t0 - t1).abs().max()) < 1e-3


@pytest.mark.parametrize("name", ["clip_small_2views_b64", "clip_small_2views_b16", "clip_small_2views",
                                  "clip_small_2views_nodrop", "clip_small_6crops"])
def test_clip_step_vs_reference_golden(name):
    G = load(name)
    B, ncrops = int(G["B"]), int(G["ncrops"])
    widths = [int(w) for w in G["widths"]]
    drop = "keep_t0" in G
    eng = AtstEngine("small", ncrops=ncrops, drop_path_rate=0.1 if drop else 0.0)
    eng.load_weights(O.recipe_weights("small", seed=int(G["seed_w"])))
    mels = [O.recipe_mel(B, w, seed=int(G["seed_x"]) + i) for i, w in enumerate(widths)]
    lens = [torch.from_numpy(l) for l in G["lengths"]]
    kt = ks = None
    if drop:
        kt = [torch.from_numpy(G[f"keep_t{i}"]) for i in range(len(O.group_views(widths[:2])))]
        ks = [torch.from_numpy(G[f"keep_s{i}"]) for i in range(len(O.group_views(widths)))]
    loss, std_s, std_t = eng.forward(mels, lens, None, kt, ks)
    eng.backward()
    s_out, t_out = eng.last_outputs
    rs, rt = rel(s_out.cpu().numpy()[:8], G["student_out"]), rel(t_out.cpu().numpy()[:8], G["teacher_out"])
    print(f"\n[{name}] loss {loss.item():.6f} (ref {float(G['loss']):.6f}) std_s {std_s.item():.5f}/{float(G['std_s']):.5f} "
          f"std_t {std_t.item():.5f}/{float(G['std_t']):.5f} out rel {rs:.2e} {rt:.2e}")
    assert abs(loss.item() - float(G["loss"])) < 5e-3
    assert abs(std_s.item() - float(G["std_s"])) < 2e-3 and abs(std_t.item() - float(G["std_t"])) < 2e-3
    assert rs < 2.5e-2 and rt < 2.5e-2
    tab = grad_table(eng, G)
    assert len(tab) == 146                                   # 147 student tensors, mask_embed never used (grad None)
    # gradients with no discontinuity behind them: tight
    assert tab["predictor.3.weight"][0] < 3e-2 and tab["predictor.1.weight"][0] < 3e-2
    big = B >= 16
    for k, (r, nerr, n) in tab.items():
        if k in CANCELLING:
            continue
        assert abs(nerr) < (5e-2 if big else 0.4), (k, nerr)
        # per-tensor rel-L2 vs the fp32 reference: ALL of it is ReLU-gate flips (test_gradient_gap_is_the_relu_gates; the same
        # cases agree with the gate-matched fp32 oracle to <= 9e-5 in parity mode, tests/test_precise_gpu.py);
        # measured worst tensor 0.125 (B=16), 0.19 (6 crops, B=2) -> x1.5.  2 views at B=2 (four BatchNorm rows) measured 0.8:
        # a bound above that says nothing, so only the norms are checked there.
        if big or ncrops == 6:
            assert r < (0.19 if big else 0.28), (k, r)
    for k in ("student.projector.1.running_mean", "student.projector.1.running_var", "teacher.projector.1.running_var"):
        net, which, _, buf = k.split(".")
        got = eng.bn_buffers[f"{net}.{which}"][buf].cpu()[sample_idx(4096)].numpy()
        assert rel(got, G["bn/" + k]) < 2e-2, k
    # EMA on the un-stepped student (ref: atst.py:29-34): fp32 elementwise -> tight
    eng.ema_update(0.99)
    for k in ("teacher.encoder.pos_embed", "teacher.encoder.blocks.3.mlp.fc1.weight", "teacher.projector.0.weight"):
        v = eng.param_view("teacher", k[len("teacher."):]).reshape(-1).cpu()
        assert rel(v[sample_idx(v.numel())].numpy(), G["ema/" + k]) < 1e-6, k


def test_frame_step_vs_reference_golden():
    G = load("frame_small")
    B = int(G["B"])
    eng = AtstEngine("small", frame=True)
    eng.load_weights(O.recipe_weights("small", frame=True, seed=11))
    mels = [O.recipe_mel(B, 1001, seed=21), O.recipe_mel(B, 1001, seed=22)]
    lens = [torch.from_numpy(l) for l in G["lengths"]]
    masks = [torch.from_numpy(G["mask"])] * 2
    loss, std_s, std_t = eng.forward(mels, lens, masks, [torch.from_numpy(G["keep_t0"])], [torch.from_numpy(G["keep_s0"])])
    eng.backward()
    s_out, t_out = eng.last_outputs
    assert s_out.shape[0] == int(G["M"])                                  # ragged masked&valid gather: exact row count
    rs, rt = rel(s_out.cpu().numpy()[::7], G["student_out"]), rel(t_out.cpu().numpy()[::7], G["teacher_out"])
    print(f"\n[frame] loss {loss.item():.6f} (ref {float(G['loss']):.6f}) out rel {rs:.2e} {rt:.2e}")
    assert abs(loss.item() - float(G["loss"])) < 5e-3
    assert abs(std_s.item() - float(G["std_s"])) < 2e-3 and abs(std_t.item() - float(G["std_t"])) < 2e-3
    assert rs < 2.5e-2 and rt < 2.5e-2                                     # same rows in the same (b, n) order
    tab = grad_table(eng, G)
    assert len(tab) == 146 and "encoder.mask_embed" in tab                 # frame model: mask_embed IS trained
    assert tab["predictor.3.weight"][0] < 3e-2
    for k, (r, nerr, n) in tab.items():                                    # ~650 head rows: well inside the B>=16 regime
        if k not in CANCELLING:
            assert abs(nerr) < 5e-2 and r < 7.5e-2, (k, r, nerr)          # measured worst 4.8e-2 (x1.5)


# measured (MI355X): bf16 loss 3.9e-4, outputs 9.3e-3 / 5.9e-3, gradients 1.9e-2 mean / 2.5e-2 worst ; fp8 (all 12 GEMMs e4m3) 1.4e-4, 0.123 / 0.072, 0.167 / 0.214 -- x 1.5
FRAME_BASE_TOL = {False: dict(loss=5e-3, out=1.4e-2, mean=2.9e-2, worst=3.8e-2), True: dict(loss=5e-3, out=0.19, mean=0.25, worst=0.32)}


@pytest.mark.parametrize("fp8", [False, True])
def test_frame_base_step_vs_reference_golden(fp8):
    """ATST-Frame *base* (the reference's train_base.sh recipe: FrameAST at embed_dim 768, 12 heads -- atstframe/audio_transformer.py:287-288) at depth 2,
    B = 8 ragged sequences per view, one block mask per sequence, against the imported reference (tests/golden/frame_base_depth2.npz): exact masked-row
    count, loss, head outputs in (b, n) row order, gradients.  fp8: the same step twice on the fp8 engine -- step 1 e4m3 forward + recording bf16
    backward, step 2 all 12 GEMMs of a block on e4m3 operands with `fp8_lean` (no bf16 LayerNorm / GELU copies) under a RAGGED head row count
    (~2.6 k rows, never a multiple of 256) -- against the fp32 reference: the e4m3 staircase, bounds = measured x 1.5."""
    G = load("frame_base_depth2")
    B, depth = int(G["B"]), int(G["depth"])
    eng = AtstEngine("base", frame=True, depth=depth, fp8=fp8)
    eng.load_weights(O.recipe_weights("base", depth=depth, frame=True, seed=71))
    mels = [O.recipe_mel(B, 1001, seed=73), O.recipe_mel(B, 1001, seed=74)]
    lens = [torch.from_numpy(l) for l in G["lengths"]]
    masks = [torch.from_numpy(G["mask"])] * 2
    tol = FRAME_BASE_TOL[fp8]
    for step in range(2 if fp8 else 1):
        if fp8:
            assert eng.fp8_bwd_state == step + 1
        loss, std_s, std_t = eng.forward(mels, lens, masks, [torch.from_numpy(G["keep_t0"])], [torch.from_numpy(G["keep_s0"])])
        eng.backward()
        s_out, t_out = eng.last_outputs
        assert s_out.shape[0] == int(G["M"])                              # ragged masked & valid gather: exact row count
        rs, rt = rel(s_out.cpu().numpy()[::23], G["student_out"]), rel(t_out.cpu().numpy()[::23], G["teacher_out"])
        tab = grad_table(eng, G)
        assert len(tab) == 14 + 11 * depth and "encoder.mask_embed" in tab          # frame model: mask_embed IS trained (146 tensors at depth 12)
        mean = sum(r * n for k, (r, _, n) in tab.items() if k not in CANCELLING) / sum(n for k, (_, _, n) in tab.items() if k not in CANCELLING)
        worst = max(((k, v) for k, v in tab.items() if k not in CANCELLING), key=lambda kv: kv[1][0])
        print(f"\n[frame base {'fp8 step %d' % (step + 1) if fp8 else 'bf16'}] rows {s_out.shape[0]} loss {loss.item():.6f} (ref {float(G['loss']):.6f}) "
              f"out rel {rs:.2e} {rt:.2e}; gradient weighted mean rel-L2 {mean:.3e}; worst {worst[0]} {worst[1][0]:.3e}")
        assert abs(loss.item() - float(G["loss"])) < tol["loss"]
        assert abs(std_s.item() - float(G["std_s"])) < 10 * tol["loss"] and abs(std_t.item() - float(G["std_t"])) < 10 * tol["loss"]
        assert rs < tol["out"] and rt < tol["out"]
        assert mean < tol["mean"] and worst[1][0] < tol["worst"], (mean, worst)
    if fp8:
        assert sum(eng.fp8_saturation().values()) == 0


def test_frame_small_fp8_step_vs_reference_golden():
    """ATST-Frame small on the fp8 engine (round 6: the all-e4m3 backward exists at d = 384) against the reference golden frame_small (12 layers, B = 2,
    ~650 masked head rows): step 1 e4m3 forward + recording backward, step 2 everything on e4m3 with fp8_lean.  Bounds = measured x 1.5."""
    G = load("frame_small")
    B = int(G["B"])
    eng = AtstEngine("small", frame=True, fp8=True)
    eng.load_weights(O.recipe_weights("small", frame=True, seed=11))
    mels = [O.recipe_mel(B, 1001, seed=21), O.recipe_mel(B, 1001, seed=22)]
    lens = [torch.from_numpy(l) for l in G["lengths"]]
    masks = [torch.from_numpy(G["mask"])] * 2
    for step in range(2):
        assert eng.fp8_bwd_state == step + 1
        loss, std_s, std_t = eng.forward(mels, lens, masks, [torch.from_numpy(G["keep_t0"])], [torch.from_numpy(G["keep_s0"])])
        eng.backward()
        s_out, t_out = eng.last_outputs
        assert s_out.shape[0] == int(G["M"])
        rs, rt = rel(s_out.cpu().numpy()[::7], G["student_out"]), rel(t_out.cpu().numpy()[::7], G["teacher_out"])
        tab = grad_table(eng, G)
        keep = {k: v for k, v in tab.items() if k not in CANCELLING}
        mean = sum(r * n for r, _, n in keep.values()) / sum(n for _, _, n in keep.values())
        worst = max(keep.items(), key=lambda kv: kv[1][0])
        print(f"\n[frame small fp8 step {step + 1}] loss {loss.item():.6f} (ref {float(G['loss']):.6f}) out rel {rs:.2e} {rt:.2e}; gradient weighted mean rel-L2 {mean:.3e}; worst {worst[0]} {worst[1][0]:.3e}")
        assert abs(loss.item() - float(G["loss"])) < FRAME_S8["loss"] and rs < FRAME_S8["out"] and rt < FRAME_S8["out"]
        assert mean < FRAME_S8["mean"] and worst[1][0] < FRAME_S8["worst"], (mean, worst)
    assert sum(eng.fp8_saturation().values()) == 0


FRAME_S8 = dict(loss=1e-2, out=0.32, mean=0.40, worst=0.53)   # measured: loss 6.7e-3, head outputs 0.21 / 0.098, gradients 0.268 mean / 0.351 worst -- x1.5.  The encoder alone is 8.6e-2 from fp32 and 5.2e-2 from
                                                              # the oracle's e4m3 emulation, which is itself 8.5e-2 from fp32 (tools/debug/frame_fp8_check.py): the heads' BatchNorm over ~650 rows + ReLU gates amplify it


@pytest.mark.parametrize("width,B", [(401, 8), (501, 32)])           # 100 tokens in 128-row tiles, S = 16 ; 125 tokens, S = 64: both PACKED (row stride < tile rows)
def test_frame_short_crop_packed_vs_oracle(width, B):
    """ATST-Frame on short crops (--anchor_len 4 / 5): the sequences are stored packed (row stride = token count), so the mask-token
    flags, the ragged row gather and its scatter must all use that stride (round-3 ADVICE: they used the padded count).  Against the
    CPU oracle (pinned to the reference by the frame goldens at full length): exact row count, loss, head outputs, gradients."""
    from parity_helpers import oracle_grads
    rng = np.random.default_rng(width)
    n_tok = (width - width % 4) // 4
    W = O.recipe_weights("small", frame=True, seed=17)
    mels = [O.recipe_mel(B, width, seed=41), O.recipe_mel(B, width, seed=42)]
    ln = torch.from_numpy(rng.integers(width // 2, width + 1, size=B))       # ragged, the same for both views (they share the mask, and ByolLoss
    ln[0] = width                                                            # chunks the rows per view: equal row counts per view)
    lens = [ln, ln]
    mask = torch.from_numpy(rng.random((B, n_tok)) < 0.6)
    masks = [mask, mask]
    keep_t = [torch.from_numpy((rng.random((12, 2, 2 * B)) < 0.95).astype(np.float32))]
    keep_s = [torch.from_numpy((rng.random((12, 2, 2 * B)) < 0.95).astype(np.float32))]
    keep_t[0][0] = 1.0; keep_s[0][0] = 1.0                                 # block 0 has DropPath rate 0: the reference never drops there (transformer.py:48-56)
    eng = AtstEngine("small", frame=True)
    eng.load_weights(W)
    loss, std_s, std_t = eng.forward(mels, lens, masks, keep_t, keep_s)
    ep = eng._student_groups[0][0]
    assert ep.RS == n_tok and ep.NP == 128 and ep.RS < ep.NP               # the packed layout is what runs
    eng.backward()
    s_out, t_out = eng.last_outputs
    fwd = lambda Wl: O.frame_atst_forward(Wl, mels, lens, masks, "small", keep_t, keep_s)
    with torch.no_grad():
        t_ref = O.frame_net_forward(W, "teacher.", mels, lens, masks, False, "small", False, keep_t, None, 0.1)
    l32, g32 = oracle_grads(W, fwd, False)
    assert s_out.shape[0] == t_ref.shape[0] == t_out.shape[0]             # ragged masked & valid gather: exact row count
    rt = rel(t_out.cpu().numpy(), t_ref.numpy())
    print(f"\n[frame packed {width}] rows {s_out.shape[0]} loss {loss.item():.6f} (oracle {l32:.6f}) teacher out rel {rt:.2e}")
    assert abs(loss.item() - l32) < 5e-3 and rt < 2.5e-2                   # same rows in the same (b, n) order
    gh = {k: eng.param_view("student", k, grad=True).detach().cpu() for k in eng.layout.entries}
    assert float(gh["encoder.mask_embed"].abs().max()) > 0.0
    bad = []
    for k, g in g32.items():
        if k in CANCELLING:
            continue
        r = rel(gh[k].reshape(-1).numpy(), g.reshape(-1).numpy())
        if r > 0.12:                                                       # ReLU-gate regime of the full-length frame goldens (measured there 4.8e-2 / 8e-2)
            bad.append((k, r))
    assert not bad, bad[:5]


def test_frame_asymmetric_step_vs_reference_golden():
    """FrameATST(symmetric=False) (methods/atstframe/model.py:73-76): teacher on the clean view 0, student on the masked view 1."""
    from audiossl_amd.models.atst import FrameATST
    G = load("frame_small_asym")
    B = int(G["B"])
    model = FrameATST("small", symmetric=False)
    eng = model.engine
    eng.load_weights(O.recipe_weights("small", frame=True, seed=13))
    mels = [O.recipe_mel(B, 1001, seed=31), O.recipe_mel(B, 1001, seed=32)]
    lens = [torch.from_numpy(l) for l in G["lengths"]]
    masks = [torch.from_numpy(G["mask"])] * 2
    loss, std_s, std_t = eng.forward(mels, lens, masks, [torch.from_numpy(G["keep_t0"])], [torch.from_numpy(G["keep_s0"])])
    eng.backward()
    s_out, t_out = eng.last_outputs
    assert s_out.shape[0] == int(G["M"]) == t_out.shape[0]                 # same masked & valid rows for both networks
    rs, rt = rel(s_out.cpu().numpy()[::7], G["student_out"]), rel(t_out.cpu().numpy()[::7], G["teacher_out"])
    print(f"\n[frame asym] loss {loss.item():.6f} (ref {float(G['loss']):.6f}) out rel {rs:.2e} {rt:.2e}")
    assert abs(loss.item() - float(G["loss"])) < 5e-3
    assert abs(std_s.item() - float(G["std_s"])) < 2e-3 and abs(std_t.item() - float(G["std_t"])) < 2e-3
    assert rs < 2.5e-2 and rt < 2.5e-2
    tab = grad_table(eng, G)
    assert len(tab) == 146 and tab["predictor.3.weight"][0] < 3e-2
    for k, (r, nerr, n) in tab.items():
        if k not in CANCELLING:
            assert abs(nerr) < 5e-2 and r < 0.12, (k, r, nerr)            # ReLU-gate flips behind ~650 BatchNorm rows (cf. symmetric: 4.8e-2)


@pytest.mark.parametrize("name,tol", [("clip_small_2views_b16", dict(g32=(1.9e-2, 2.7e-2), emu=(1.7e-2, 2.8e-2), flip=5e-2)),
                                      ("frame_small", dict(g32=(2.6e-2, 3.2e-2), emu=(6.5e-3, 9e-3), flip=2e-2))])
def test_gradient_gap_is_the_relu_gates(name, tol):
    """Pins the HIP-vs-reference gradient gap to its cause (measurements: profiles/r02_parity_emu.txt, tools/parity_emu.py).
    The fp32 oracle evaluated with the HIP run's ReLU gate patterns (the step's only discontinuity, behind BatchNorm):
      * moves away from the plain fp32 oracle by the whole gap (10.6 % clip B=16, 3.7 % frame) -- the gates ARE the gap;
      * agrees with HIP to 1.3 % / 1.7 % (encoder mean), and the bf16-emulating oracle (rounds where HIP rounds) to
        1.1 % / 0.40 %: the smooth part of the error.  Tolerances = measured x 1.5 (mean, worst tensor)."""
    from parity_helpers import step_three_ways
    losses, tab = step_three_ways(name, with_plain=False)
    assert abs(losses["hip"] - losses["fp32"]) < 5e-3 and abs(losses["fp32"] - losses["golden"]) < 1e-5
    a, b, c = tab["HIP vs fp32+gates"], tab["HIP vs emulated+gates"], tab["fp32+gates vs fp32"]
    print(f"\n[{name}] HIP vs fp32 {tab['HIP vs fp32']['mean']:.3e} | vs fp32+gates {a['mean']:.3e} (worst {a['worst'][0]:.3e}) | "
          f"vs emulated+gates {b['mean']:.3e} (worst {b['worst'][0]:.3e}) | gates alone {c['mean']:.3e}")
    assert a["mean"] < tol["g32"][0] and a["worst"][0] < tol["g32"][1], a
    assert b["mean"] < tol["emu"][0] and b["worst"][0] < tol["emu"][1], b
    assert all(v < tol["g32"][1] for v in a["heads"].values()) and b["heads"]["predictor.3.weight"] < 8e-3
    assert c["mean"] > tol["flip"]                                   # the gate pattern alone reproduces the gap
    assert abs(c["mean"] - tab["HIP vs fp32"]["mean"]) < 0.25 * tab["HIP vs fp32"]["mean"]


def test_optimizer_trajectory_vs_oracle():
    """Two full steps (fwd+bwd+HF-AdamW+EMA) against the CPU oracle on the same inputs.  Adam's normalised update is
    O(lr) per element whatever |g| is, so the comparison scale is lr: |dp| <= 2.5 lr per step (sign flips of tiny g)."""
    B, lr, wd, ema = 2, 1e-3, 0.04, 0.99
    W = O.recipe_weights("small", depth=2, seed=5)
    eng = AtstEngine("small", depth=2, drop_path_rate=0.0)
    eng.load_weights(W)
    leaves = {k: v.requires_grad_(True) for k, v in W.items()
              if k.startswith("student.") and v.dtype == torch.float32 and "running" not in k}
    st = {k: (torch.zeros_like(v), torch.zeros_like(v)) for k, v in leaves.items()}
    reg, _ = O.param_groups([(k[len("student."):], tuple(v.shape)) for k, v in leaves.items()])
    for step in (1, 2):
        mels = [O.recipe_mel(B, 1001, seed=50 + step), O.recipe_mel(B, 1001, seed=60 + step)]
        lens = [torch.tensor([1001, 1001]), torch.tensor([1001, 900])]
        loss_h, _, _ = eng.forward(mels, lens)
        eng.backward()
        eng.optimizer_step(lr, wd, ema)
        loss_o, _, _ = O.atst_forward(W, mels, lens, "small", 2, depth=2, drop_path_rate=0.0)
        for v in leaves.values():
            v.grad = None
        loss_o.backward()
        assert abs(loss_h.item() - loss_o.item()) < 2e-2
        with torch.no_grad():
            for k, v in leaves.items():
                if v.grad is not None:
                    O.hf_adamw_step(v, v.grad, st[k][0], st[k][1], step, lr, wd if k[len("student."):] in reg else 0.0)
        for v in leaves.values():
            v.requires_grad_(False)
        O.ema_update(W, ema)
        for v in leaves.values():
            v.requires_grad_(True)
    worst = 0.0
    for name in eng.layout.entries:
        d = float((eng.param_view("student", name).detach().cpu() - W["student." + name].detach()).abs().max())
        worst = max(worst, d)
        assert d <= 2.5 * lr * 2, (name, d)
    assert float((eng.param_view("student", "encoder.mask_embed").cpu() - O.recipe_weights("small", depth=2, seed=5)
                  ["student.encoder.mask_embed"]).abs().max()) == 0.0      # never updated, not even weight-decayed
    for name in eng.layout.entries:
        if not name.startswith("predictor."):
            got = eng.param_view("teacher", name).detach().cpu()
            assert float((got - W["teacher." + name]).abs().max()) <= 1e-4, name
    print(f"\n[trajectory] worst |param diff| after 2 steps: {worst:.2e} (lr {lr})")


@pytest.mark.parametrize("precise", [False, True])
def test_hires_patch_geometry_encoder_vs_oracle(precise):
    """BASELINE.json configs[4] geometry: 128 mel bands, one patch row of 128 x 8 (the reference's --patch_h / --patch_w with
    spec_h = n_mels, methods/atstframe/train.py:15,50-51), 10 s @ 32 kHz = 2001 frames -> 250 patches, K = 1024 patch vectors.
    Forward CLS and the gradient of the smooth objective sum(CLS * R) against the CPU oracle's autograd: bf16 path at the bf16
    tolerances of test_base_arch_encoder_vs_oracle, parity mode at 1e-4."""
    S, depth = 5, 2
    W = O.recipe_weights("small", depth=depth, seed=41, patch_h=128, patch_w=8)
    eng = AtstEngine("small", depth=depth, patch_h=128, patch_w=8, precise=precise)
    eng.load_weights(W)
    mel = O.recipe_mel(S, 2001, seed=43, n_mels=128)
    length = torch.tensor([2001, 2001, 1555, 1281, 809])
    keep = (torch.rand(depth, 2, S, generator=torch.Generator().manual_seed(9)) > 0.3).float()
    keep[0] = 1.0
    leaves = {k: v.requires_grad_(True) for k, v in W.items() if k.startswith("student.encoder.") and v.dtype == torch.float32}
    cls_o = O.encoder_forward(W, "student.encoder.", mel, length, "small", depth, keep=keep)
    R = torch.from_numpy(np.random.default_rng(45).standard_normal((S, 384)).astype(np.float32))
    (cls_o * R).sum().backward()
    ep = eng._pass("student", S, 2001, True, 0)
    assert ep.n_tok == 250 and ep.NP == 256
    valid = eng._valid(length, 1)
    assert torch.equal(valid.cpu() - 1, O.patch_length(length, 128, 128, 8).int())         # integer: bit-exact
    out = ep.forward(mel.cuda(), valid, None, eng.drop_path_scales(S, keep))
    cls = out.float().reshape(S, 256, 384)[:, 0].cpu()
    assert rel(cls.numpy(), cls_o.detach().numpy()) < (2e-5 if precise else 1e-2)
    eng.g32.zero_(); ep.dout.zero_()
    rows = (torch.arange(S, dtype=torch.int32, device="cuda") * 256).contiguous()
    if precise:
        ep.dout.index_copy_(0, rows.long(), R.cuda())
    else:
        hip.call("atst_scatter_rows_bf16", hip.ptr(R.cuda()), hip.ptr(rows), S, 384, hip.ptr(ep.dout), hip.stream())
    ep.backward()
    num = den = 0.0
    worst = ("", 0.0)
    for name in eng.layout.entries:
        if not name.startswith("encoder.") or name == "encoder.mask_embed":
            continue
        g = eng.param_view("student", name, grad=True).double().cpu()
        go = leaves["student." + name].grad.double()
        r = float((g - go).norm() / (go.norm() + 1e-30))
        num += r * g.numel(); den += g.numel()
        if r > worst[1]:
            worst = (name, r)
    print(f"\n[hires 128x8 encoder grad, precise={precise}] weighted mean rel-L2 {num / den:.3e}; worst {worst[0]} {worst[1]:.3e}")
    if precise:
        assert worst[1] < 1e-4
    else:
        assert num / den < 1.0e-2 and worst[1] < 2.4e-2
