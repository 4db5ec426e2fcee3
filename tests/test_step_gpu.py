"""End-to-end parity of the HIP training step (bf16 MFMA compute, fp32 statistics / residual stream / master weights)
against (a) goldens produced by the upstream reference and (b) the CPU oracle, through the C ABI.

Stated tolerances (reference is fp32; the HIP path rounds GEMM operands to bf16, 2^-9 relative per element):
  loss / std monitors   : |d| <= 2e-3
  head outputs, CLS     : rel-L2 <= 2e-2
  student gradients     : per-tensor rel-L2 on the golden samples <= 6e-2, norm error <= 3e-2, and the
                          parameter-count-weighted mean rel-L2 <= 2.5e-2   (north_star asks 1e-3: NOT met in bf16;
                          see DESIGN.md "Precision")
  integer work (patch gather, valid lengths, ragged row order, zero gradient of never-used mask_embed): bit-exact.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from audiossl_amd import hip  # noqa: E402
from audiossl_amd.engine import AtstEngine  # noqa: E402
from oracle import atst_oracle as O  # noqa: E402

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    return np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False)


def sample_idx(n, k=192):
    return np.unique(np.linspace(0, n - 1, num=min(n, k)).astype(np.int64))


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


def check_grads(eng, G, prefix_map=lambda n: n, tol_each=6e-2, tol_norm=3e-2, tol_mean=2.5e-2):
    tot, acc, worst = 0.0, 0.0, ("", 0.0)
    for name, (off, shape) in eng.layout.entries.items():
        key = prefix_map(name)
        g = eng.param_view("student", name, grad=True).reshape(-1).double().cpu()
        if "gnone/" + key in G:
            assert float(g.abs().max()) == 0.0, name            # grad None in the reference <-> exactly untouched here
            continue
        gn = float(G["gnorm/" + key])
        r = rel(g[sample_idx(g.numel())].numpy(), G["gsamp/" + key])
        assert abs(float(g.norm()) - gn) <= tol_norm * gn, (name, float(g.norm()), gn)
        assert r <= tol_each, (name, r)
        n = g.numel()
        tot += n; acc += n * r
        if r > worst[1]:
            worst = (name, r)
    mean = acc / tot
    print(f"\n[grad parity] weighted mean rel-L2 {mean:.3e}; worst {worst[0]} {worst[1]:.3e}")
    assert mean <= tol_mean, mean
    return mean


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
    # rows beyond the valid length are never consumed downstream (not keys, output unused) but still match
    assert rel(b0[:, ::10, ::4], G["block0"]) < 1e-2
    assert rel(b1[:, ::10, ::4], G["block1"]) < 1e-2
    cls = out.float().reshape(S, 256, 384)[:, 0].cpu().numpy()
    assert rel(cls, G["cls"]) < 1e-2


@pytest.mark.parametrize("name", ["clip_small_2views", "clip_small_2views_nodrop", "clip_small_6crops"])
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
    print(f"\n[{name}] loss {loss.item():.6f} (ref {float(G['loss']):.6f}) std_s {std_s.item():.5f}/{float(G['std_s']):.5f} "
          f"std_t {std_t.item():.5f}/{float(G['std_t']):.5f} out rel {rel(s_out.cpu().numpy(), G['student_out']):.2e} "
          f"{rel(t_out.cpu().numpy(), G['teacher_out']):.2e}")
    assert abs(loss.item() - float(G["loss"])) < 5e-3
    assert abs(std_s.item() - float(G["std_s"])) < 2e-3 and abs(std_t.item() - float(G["std_t"])) < 2e-3
    assert rel(s_out.cpu().numpy(), G["student_out"]) < 2e-2 and rel(t_out.cpu().numpy(), G["teacher_out"]) < 2e-2
    check_grads(eng, G)
    for k in ("student.projector.1.running_mean", "student.projector.1.running_var", "teacher.projector.1.running_var"):
        net, which, _, buf = k.split(".")
        got = eng.bn_buffers[f"{net}.{which}"][buf].cpu()[sample_idx(4096)].numpy()
        assert rel(got, G["bn/" + k]) < 1e-2, k
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
    print(f"\n[frame] loss {loss.item():.6f} (ref {float(G['loss']):.6f}) out rel {rel(s_out.cpu().numpy()[::7], G['student_out']):.2e}")
    assert abs(loss.item() - float(G["loss"])) < 5e-3
    assert abs(std_s.item() - float(G["std_s"])) < 2e-3 and abs(std_t.item() - float(G["std_t"])) < 2e-3
    assert rel(s_out.cpu().numpy()[::7], G["student_out"]) < 2e-2          # same rows in the same (b, n) order
    assert rel(t_out.cpu().numpy()[::7], G["teacher_out"]) < 2e-2
    check_grads(eng, G)


def test_optimizer_trajectory_vs_oracle():
    """Two full steps (fwd+bwd+HF-AdamW+EMA) against the CPU oracle run on the same inputs: parameters after the
    update agree to bf16-gradient accuracy, first-step update direction is sign-exact where |g| is not tiny."""
    B = 2
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
        lr, wd, ema = 1e-3, 0.04, 0.99
        loss_h, _, _ = eng.forward(mels, lens)
        eng.backward()
        eng.optimizer_step(lr, wd, ema)
        loss_o, _, _ = O.atst_forward(W, mels, lens, "small", 2, depth=2, drop_path_rate=0.0)
        for v in leaves.values():
            v.grad = None
        loss_o.backward()
        assert abs(loss_h.item() - loss_o.item()) < 3e-3
        with torch.no_grad():
            for k, v in leaves.items():
                if v.grad is None:
                    continue
                O.hf_adamw_step(v, v.grad, st[k][0], st[k][1], step, lr, wd if k[len("student."):] in reg else 0.0)
        for v in W.values():
            v.requires_grad_(False) if v.dtype == torch.float32 else None
        O.ema_update(W, ema)
        for v in leaves.values():
            v.requires_grad_(True)
    worst = 0.0
    for name in eng.layout.entries:
        got = eng.param_view("student", name).detach().cpu()
        want = W["student." + name].detach()
        # Adam's normalised update is O(lr) per element whatever |g| is, so compare against that scale
        d = float((got - want).abs().max())
        worst = max(worst, d)
        assert d <= 2.5 * 1e-3 * 2, (name, d)
    for name in eng.layout.entries:
        if name.startswith("predictor."):
            continue
        got = eng.param_view("teacher", name).detach().cpu()
        assert float((got - W["teacher." + name]).abs().max()) <= 1e-4, name
    print(f"\n[trajectory] worst |param diff| after 2 steps: {worst:.2e}")
