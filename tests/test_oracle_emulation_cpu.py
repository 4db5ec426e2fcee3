"""CPU checks of the oracle's emulation contexts (test infrastructure for the GPU parity tests: tests/test_ops_gpu.py::test_fp8_dgrad_step_base and
friends inject HIP's scales into them).  Small geometry, seconds on one core: what the contexts do must hold without a GPU."""
import contextlib
import math

import torch

from oracle import atst_oracle as O

DEPTH, B = 2, 4


def _setup():
    W = O.recipe_weights("base", depth=DEPTH, seed=7)
    mels = [O.recipe_mel(B, 201, seed=1), O.recipe_mel(B, 201, seed=2)]            # 2 s crops: 50 tokens
    lens = [torch.full((B,), 201)] * 2
    fwd = lambda Wl: O.atst_forward(Wl, mels, lens, "base", 2, depth=DEPTH, drop_path_rate=0.0)
    return W, fwd


def _grads(W, fwd, ctxs=()):
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in W.items() if k.startswith("student.") and v.dtype == torch.float32 and "running" not in k}
    Wl = {k: (leaves[k] if k in leaves else v.clone()) for k, v in W.items()}
    with contextlib.ExitStack() as stack:
        for c in ctxs:
            stack.enter_context(c)
        loss = fwd(Wl)[0]
        loss.backward()
    return float(loss.detach()), {k[len("student."):]: v.grad.detach() for k, v in leaves.items() if v.grad is not None}


def _rel(a, b):
    return float((a - b).norm() / (b.norm() + 1e-30))


def test_fp8_dgrad_emulation_sites_and_scales():
    torch.manual_seed(0)
    W, fwd = _setup()
    sites = ("mlp.fc2.weight", "mlp.fc1.weight", "attn.proj.weight", "attn.qkv.weight")
    rec3, rec4 = O.emulate_fp8_dgrad(None), O.emulate_fp8_dgrad(None, qkv=True)
    l3, g3 = _grads(W, fwd, (O.emulate_bf16(), O.emulate_fp8(), rec3))
    l4, g4 = _grads(W, fwd, (O.emulate_bf16(), O.emulate_fp8(), rec4))
    # recording changes nothing, and the qkv site is recorded only when asked for
    assert l3 == l4 and all(torch.equal(g3[k], g4[k]) for k in g3)
    assert len(rec3.amax) == 3 * DEPTH and len(rec4.amax) == 4 * DEPTH
    assert all(k.endswith(sites[:3]) for k in rec3.amax) and any(k.endswith(sites[3]) for k in rec4.amax)
    sc = rec4.next_scales()
    assert all(abs(sc[k] - 448.0 / (2.0 * rec4.amax[k])) < 1e-6 * sc[k] for k in sc) and all(math.isfinite(v) and v > 0 for v in sc.values())
    # e4m3 dgrads on the three MLP / proj sites: the gradients move by the e4m3 staircase, a few per cent
    sc3 = {k: v for k, v in sc.items() if not k.endswith(sites[3])}
    _, g_d3 = _grads(W, fwd, (O.emulate_bf16(), O.emulate_fp8(), O.emulate_fp8_dgrad(sc3)))
    _, g_d4 = _grads(W, fwd, (O.emulate_bf16(), O.emulate_fp8(), O.emulate_fp8_dgrad(sc, qkv=True)))
    _, g_w4 = _grads(W, fwd, (O.emulate_bf16(), O.emulate_fp8(), O.emulate_fp8_dgrad(sc, wgrad=True, qkv=True)))
    last = f"encoder.blocks.{DEPTH - 1}."
    for nm in ("mlp.fc1.weight", "attn.qkv.weight"):
        k0 = "encoder.blocks.0." + nm
        assert 1e-3 < _rel(g_d3[k0], g3[k0]) < 0.3, nm                 # block 0 sits behind block 1's e4m3 dgrads
    # the last block's fc2 weight gradient has the loss gradient itself as dY (no e4m3 dgrad upstream): it changes only when the weight gradient
    # itself goes e4m3; every weight gradient moves by a few per cent of e4m3 staircase then
    k = last + "mlp.fc2.weight"
    assert torch.equal(g_d4[k], g3[k]) and torch.equal(g_d3[k], g3[k])
    for nm in sites:
        r = _rel(g_w4[last + nm], g_d4[last + nm])
        assert 1e-3 < r < 0.2, (nm, r)
    # the qkv site off: the qkv weight gradient stays on bf16 operands even with wgrad=True (same dgrads: bit-identical), the others do not
    _, g_w3 = _grads(W, fwd, (O.emulate_bf16(), O.emulate_fp8(), O.emulate_fp8_dgrad(sc3, wgrad=True)))
    assert torch.equal(g_w3[last + "attn.qkv.weight"], g_d3[last + "attn.qkv.weight"])
    assert not torch.equal(g_w3[last + "mlp.fc1.weight"], g_d3[last + "mlp.fc1.weight"])
    # the qkv dgrad in e4m3 moves block 0's gradients beyond what the three-site emulation gives
    assert _rel(g_d4["encoder.blocks.0.mlp.fc1.weight"], g_d3["encoder.blocks.0.mlp.fc1.weight"]) > 1e-4


def test_bf16_emulation_sites_switch_off_exactly():
    """Encoder-only objective sum(CLS * R) (no head behind it: the heads have rounding of their own): with every encoder site switched off the
    emulation IS the fp32 oracle (same forward bit for bit, gradients to fp32 rounding); with all of them on the gradient moves by bf16-sized amounts (tools/rounding_sites.py builds on this)."""
    W, _ = _setup()
    mel, lens = O.recipe_mel(B, 201, seed=1), torch.full((B,), 201)
    R = torch.randn(B, 768, generator=torch.Generator().manual_seed(3))
    fwd = lambda Wl: ((O.encoder_forward(Wl, "student.encoder.", mel, lens, "base", depth=DEPTH, drop_path_rate=0.0) * R).sum(),)
    l32, g32 = _grads(W, fwd)
    l16, g16 = _grads(W, fwd, (O.emulate_bf16(),))
    loff, goff = _grads(W, fwd, (O.emulate_bf16(True, off=O.ROUNDING_SITES),))
    assert loff == l32 and max(_rel(goff[k], g32[k]) for k in g32) < 1e-5        # (its GELU is its own autograd function: fp32 rounding apart)
    k = "encoder.blocks.0.attn.qkv.weight"
    assert 1e-4 < _rel(g16[k], g32[k]) < 5e-2
    # one site at a time: switching off the weight shadows (the one large site, profiles/r05_rounding_sites.txt) brings the gradient closer to fp32
    _, gw = _grads(W, fwd, (O.emulate_bf16(True, off=("w",)),))
    assert _rel(gw[k], g32[k]) < _rel(g16[k], g32[k])
