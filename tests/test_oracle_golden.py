"""Pin the CPU oracle (oracle/atst_oracle.py) against goldens produced by importing the upstream reference
(tests/golden/make_golden.py).  fp32 tolerances: <=1e-4 rel for outputs/loss, <=1e-3 rel for gradients
(SURVEY.md section 7 step 2); integer work (patch gather, patch_length, mask) is bit-exact."""
import os

import numpy as np
import pytest
import torch

from oracle import atst_oracle as O

torch.set_num_threads(8)


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"), allow_pickle=False)


def sample_idx(n, k=192):
    return np.unique(np.linspace(0, n - 1, num=min(n, k)).astype(np.int64))


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


def grad_check(G, named, tol_norm=1e-3, tol_samp=2e-3):
    worst = 0.0
    for name, p in named:
        if "gnone/" + name in G:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, name
            continue
        g = p.grad.reshape(-1).double()
        gn = float(G["gnorm/" + name])
        assert abs(float(g.norm()) - gn) <= tol_norm * gn + 1e-12, (name, float(g.norm()), gn)
        r = rel(g[sample_idx(g.numel())].numpy(), G["gsamp/" + name])
        worst = max(worst, r)
        assert r <= tol_samp, (name, r)
    return worst


def student_leaves(W):
    leaves = []
    for k, v in W.items():
        if k.startswith("student.") and v.dtype == torch.float32 and not k.endswith(("running_mean", "running_var")):
            v.requires_grad_(True)
            leaves.append((k[len("student."):], v))
    return leaves


def test_patch_gather_bit_exact(golden_dir):
    G = load(golden_dir, "clip_depth2_blocks")
    S = int(G["S"])
    probe = (torch.arange(S)[:, None, None, None] * 100000 + torch.arange(64)[None, None, :, None] * 1001
             + torch.arange(1001)[None, None, None, :]).float()
    p = O.patchify(probe).numpy().astype(np.int32)[:, ::25]
    assert np.array_equal(p, G["patches_probe"])
    assert np.array_equal(O.patch_length(torch.from_numpy(G["length"])).numpy(), G["patch_length"])


def test_depth2_blocks(golden_dir):
    G = load(golden_dir, "clip_depth2_blocks")
    W = O.recipe_weights("small", depth=2, seed=3)
    mel = O.recipe_mel(int(G["S"]), 1001, seed=5)
    length = torch.from_numpy(G["length"])
    x, plen = O.encoder_tokens(W, "student.encoder.", mel, length, True)
    assert rel(x.numpy()[:, ::10, ::4], G["tokens"]) < 1e-6
    bias = O.key_padding_bias(251, plen + 1)
    assert np.array_equal(bias[:, 0, 0, :].numpy(), G["attn_mask_row"])
    cls, blocks = O.encoder_forward(W, "student.encoder.", mel, length, depth=2, drop_path_rate=0.0, return_blocks=True)
    assert rel(blocks[0].numpy()[:, ::10, ::4], G["block0"]) < 1e-5
    assert rel(blocks[1].numpy()[:, ::10, ::4], G["block1"]) < 1e-5
    assert rel(cls.numpy(), G["cls"]) < 1e-5


@pytest.mark.parametrize("name", ["clip_small_2views", "clip_small_2views_nodrop", "clip_small_6crops", "clip_small_2views_b16"])
def test_clip_step(golden_dir, name):
    G = load(golden_dir, name)
    B, ncrops = int(G["B"]), int(G["ncrops"])
    widths = [int(w) for w in G["widths"]]
    W = O.recipe_weights("small", seed=int(G["seed_w"]))
    leaves = student_leaves(W)
    mels = [O.recipe_mel(B, w, seed=int(G["seed_x"]) + i) for i, w in enumerate(widths)]
    lens = [torch.from_numpy(l) for l in G["lengths"]]
    kt = ks = None
    if "keep_t0" in G:
        kt = [torch.from_numpy(G[f"keep_t{i}"]) for i in range(len(O.group_views(widths[:2])))]
        ks = [torch.from_numpy(G[f"keep_s{i}"]) for i in range(len(O.group_views(widths)))]
    loss, std_s, std_t = O.atst_forward(W, mels, lens, "small", ncrops, kt, ks)
    loss.backward()
    assert abs(loss.item() - float(G["loss"])) < 1e-5 * max(1.0, abs(float(G["loss"])))
    assert abs(std_s.item() - float(G["std_s"])) < 1e-5
    assert abs(std_t.item() - float(G["std_t"])) < 1e-5
    grad_check(G, [(n, p) for n, p in leaves])
    for k in ("student.projector.1.running_mean", "student.projector.1.running_var", "teacher.projector.1.running_var"):
        assert rel(W[k].detach()[sample_idx(4096)].numpy(), G["bn/" + k]) < 1e-4, k
    for k, v in W.items():
        v.requires_grad_(False) if v.dtype == torch.float32 else None
    O.ema_update(W, 0.99)
    for k in ("teacher.encoder.pos_embed", "teacher.encoder.blocks.3.mlp.fc1.weight", "teacher.projector.0.weight",
              "teacher.projector.1.running_var"):
        v = W[k].reshape(-1)
        assert rel(v[sample_idx(v.numel())].numpy(), G["ema/" + k]) < 1e-6, k


def test_encoder_grad(golden_dir):
    """Well-conditioned encoder-only pin: L = sum(CLS * R), ragged lengths, DropPath injected."""
    G = load(golden_dir, "clip_encoder_grad")
    S = int(G["S"])
    W = O.recipe_weights("small", seed=21)
    leaves = [(k[len("student.encoder."):], v.requires_grad_(True)) for k, v in W.items() if k.startswith("student.encoder.")]
    R = torch.from_numpy(np.random.default_rng(29).standard_normal((S, 384)).astype(np.float32))
    cls = O.encoder_forward(W, "student.encoder.", O.recipe_mel(S, 1001, seed=23), torch.from_numpy(G["length"]), "small",
                            keep=torch.from_numpy(G["keep"]))
    (cls * R).sum().backward()
    assert rel(cls.detach().numpy(), G["cls"]) < 1e-5
    grad_check(G, leaves)


def test_frame_step(golden_dir):
    G = load(golden_dir, "frame_small")
    B = int(G["B"])
    W = O.recipe_weights("small", frame=True, seed=11)
    leaves = student_leaves(W)
    mels = [O.recipe_mel(B, 1001, seed=21), O.recipe_mel(B, 1001, seed=22)]
    lens = [torch.from_numpy(l) for l in G["lengths"]]
    # restated block-mask sampler reproduces the committed mask from the same numpy stream
    rs = np.random.RandomState(99)
    m = np.stack([O.block_mask(250, 0.65, 5, rng=rs) for _ in range(B)])
    assert np.array_equal(m, G["mask"])
    masks = [torch.from_numpy(m)] * 2
    loss, std_s, std_t = O.frame_atst_forward(W, mels, lens, masks, "small",
                                              [torch.from_numpy(G["keep_t0"])], [torch.from_numpy(G["keep_s0"])])
    loss.backward()
    assert abs(loss.item() - float(G["loss"])) < 1e-5
    assert abs(std_s.item() - float(G["std_s"])) < 1e-5 and abs(std_t.item() - float(G["std_t"])) < 1e-5
    grad_check(G, [(n, p) for n, p in leaves])


def test_frame_base_step(golden_dir):
    """ATST-Frame *base* (FrameAST at embed_dim 768, 12 heads: atstframe/audio_transformer.py:287-288, train_base.sh) at depth 2, B = 8 ragged sequences
    per view, one block mask per sequence: the oracle against the imported reference's MultiCropWrapper + ByolLoss (make_golden.py frame_base)."""
    G = load(golden_dir, "frame_base_depth2")
    B, depth = int(G["B"]), int(G["depth"])
    W = O.recipe_weights("base", depth=depth, frame=True, seed=71)
    leaves = student_leaves(W)
    mels = [O.recipe_mel(B, 1001, seed=73), O.recipe_mel(B, 1001, seed=74)]
    lens = [torch.from_numpy(l) for l in G["lengths"]]
    rs = np.random.RandomState(77)
    m = np.stack([O.block_mask(250, 0.65, 5, rng=rs) for _ in range(B)])
    assert np.array_equal(m, G["mask"])
    masks = [torch.from_numpy(m)] * 2
    loss, std_s, std_t = O.frame_atst_forward(W, mels, lens, masks, "base", [torch.from_numpy(G["keep_t0"])], [torch.from_numpy(G["keep_s0"])], depth=depth)
    loss.backward()
    assert abs(loss.item() - float(G["loss"])) < 1e-5
    assert abs(std_s.item() - float(G["std_s"])) < 1e-5 and abs(std_t.item() - float(G["std_t"])) < 1e-5
    grad_check(G, [(n, p) for n, p in leaves])


def test_frame_step_asymmetric(golden_dir):
    """FrameATST(symmetric=False): teacher on view 0, student on the masked view 1, one cosine pair (model.py:73-76)."""
    G = load(golden_dir, "frame_small_asym")
    B = int(G["B"])
    W = O.recipe_weights("small", frame=True, seed=13)
    leaves = student_leaves(W)
    mels = [O.recipe_mel(B, 1001, seed=31), O.recipe_mel(B, 1001, seed=32)]
    lens = [torch.from_numpy(l) for l in G["lengths"]]
    masks = [torch.from_numpy(G["mask"])] * 2
    loss, std_s, std_t = O.frame_atst_forward(W, mels, lens, masks, "small", [torch.from_numpy(G["keep_t0"])],
                                              [torch.from_numpy(G["keep_s0"])], symmetric=False)
    loss.backward()
    assert abs(loss.item() - float(G["loss"])) < 1e-5
    assert abs(std_s.item() - float(G["std_s"])) < 1e-5 and abs(std_t.item() - float(G["std_t"])) < 1e-5
    grad_check(G, [(n, p) for n, p in leaves])


def test_schedules_and_groups(golden_dir):
    G = load(golden_dir, "schedules")
    idx = G["idx"]
    lr = O.cosine_scheduler_step(5e-4 * 4 * 384 / 256, 1e-6, 39100, 1300)
    wd = O.cosine_scheduler_step(0.04, 0.4, 39100, 0)
    ema = O.cosine_scheduler_step(0.99, 1, 39100, 0)
    assert len(lr) == int(G["lr_len"])
    assert np.array_equal(lr[idx], G["lr"]) and np.array_equal(wd[idx], G["wd"]) and np.array_equal(ema[idx], G["ema"])
    shapes = [(k, s) for k, s in O.student_shapes("small")
              if not k.endswith(("running_mean", "running_var", "num_batches_tracked"))]
    assert [k for k, _ in shapes] == list(G["param_names"])
    reg, noreg = O.param_groups(shapes)
    assert reg == list(G["reg"]) and noreg == list(G["noreg"])
    W = O.recipe_weights("small")
    assert list(W.keys()) == list(G["state_dict_keys"])
    n_student = sum(int(np.prod(s)) for _, s in shapes)
    assert n_student == int(G["n_student"]) == 26211328


def test_augmentations_vs_reference_golden():
    """RandomResizeCrop / Mixup restatements against the reference's byol_a.py run with replayed draws."""
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "aug_byol_a.npz"))
    for k, W in enumerate(G["rrc_widths"]):
        x = O.recipe_mel(1, int(W), seed=200 + k)[0]
        i, j, h, w = (int(v) for v in G["rrc_params"][k])
        y = O.random_resize_crop(x, i, j, h, w)
        assert np.abs(y[0, ::2, ::3].numpy() - G[f"rrc_out{k}"]).max() < 1e-6
    for k, (Wx, Wz, idx, start) in enumerate(G["mix_meta"]):
        z = O.recipe_mel(1, int(Wz), seed=300 + k)[0]; x = O.recipe_mel(1, int(Wx), seed=310 + k)[0]
        y = O.log_mixup_exp(x, z, float(G[f"mix_alpha{k}"]), int(start))
        assert np.abs(y[0, ::2, ::3].numpy() - G[f"mix_out{k}"]).max() < 1e-5


def test_frame_inference_api_vs_reference_golden():
    """FrameAST.get_intermediate_layers (scene / timestamp embeddings of atstframe/embedding.py) restated by the oracle."""
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "frame_inference_api.npz"))
    W = O.recipe_weights("small", frame=True, seed=61)
    x, length = O.recipe_mel(3, 1001, seed=63), torch.from_numpy(G["length"])
    scene = O.frame_intermediate_layers(W, "teacher.encoder.", x, length, n=3, scene=True)
    frames = O.frame_intermediate_layers(W, "teacher.encoder.", x, length, n=2, scene=False)
    assert tuple(frames.shape) == tuple(G["frames_shape"])
    assert np.abs(scene.numpy() - G["scene"]).max() < 2e-5
    assert np.abs(frames.numpy()[:, ::5, ::4] - G["frames"]).max() < 5e-5
    short = O.frame_intermediate_layers(W, "teacher.encoder.", O.recipe_mel(2, 401, seed=65), torch.tensor([401, 401]), n=12)
    assert np.abs(short.numpy() - G["scene_short"]).max() < 2e-5


def test_base_encoder_grad_vs_reference_golden(golden_dir):
    """ATST-base geometry (d = 768, 12 heads; what AST_base builds, audio_transformer.py:371-374) at depth 3 pinned to the imported
    reference: per-block activations, CLS and the gradient of sum(CLS * R), ragged lengths, recorded DropPath draws."""
    G = load(golden_dir, "base_depth3_encoder_grad")
    S, depth = int(G["S"]), int(G["depth"])
    W = O.recipe_weights("base", depth=depth, seed=31)
    leaves = [(k[len("student.encoder."):], v.requires_grad_(True)) for k, v in W.items() if k.startswith("student.encoder.")]
    length = torch.from_numpy(G["length"])
    assert np.array_equal(O.patch_length(length).numpy(), G["patch_length"])
    assert np.array_equal(np.array(O.drop_path_rates(depth), np.float64), G["rates"])
    R = torch.from_numpy(np.random.default_rng(35).standard_normal((S, 768)).astype(np.float32))
    cls, blocks = O.encoder_forward(W, "student.encoder.", O.recipe_mel(S, 1001, seed=33), length, "base", depth,
                                    keep=torch.from_numpy(G["keep"]), return_blocks=True)
    (cls * R).sum().backward()
    for i, b in enumerate(blocks):
        assert rel(b.detach().numpy()[:, ::10, ::8], G[f"block{i}"]) < 1e-5, i
    assert rel(cls.detach().numpy(), G["cls"]) < 1e-5
    grad_check(G, leaves)


def test_base_step_vs_reference_golden(golden_dir):
    """2-view training step of an ATST("base")-shaped model at depth 2 (reference MultiCropWrapper + ByolLoss around AST(768, 12 heads))."""
    G = load(golden_dir, "base_2views_depth2")
    B, depth = int(G["B"]), int(G["depth"])
    W = O.recipe_weights("base", depth=depth, seed=41)
    leaves = student_leaves(W)
    mels = [O.recipe_mel(B, int(w), seed=43 + i) for i, w in enumerate(G["widths"])]
    lens = [torch.from_numpy(l) for l in G["lengths"]]
    with torch.no_grad():
        t = O.net_forward(W, "teacher.", mels, lens, "base", False, [torch.from_numpy(G["keep_t0"])], depth, 0.1)
    s = O.net_forward(W, "student.", mels, lens, "base", True, [torch.from_numpy(G["keep_s0"])], depth, 0.1)
    loss, std_s, std_t = O.byol_loss(s, t, 2)
    loss.backward()
    assert rel(t.numpy()[:8], G["teacher_out"]) < 1e-5 and rel(s.detach().numpy()[:8], G["student_out"]) < 1e-5
    assert abs(loss.item() - float(G["loss"])) < 1e-5 * max(1.0, abs(float(G["loss"])))
    assert abs(std_s.item() - float(G["std_s"])) < 1e-5 and abs(std_t.item() - float(G["std_t"])) < 1e-5
    grad_check(G, [(n, p) for n, p in leaves])
    for k in ("student.projector.1.running_mean", "student.projector.1.running_var", "teacher.projector.1.running_var"):
        assert rel(W[k].detach()[sample_idx(4096)].numpy(), G["bn/" + k]) < 1e-4, k
    for v in W.values():
        v.requires_grad_(False) if v.dtype == torch.float32 else None
    O.ema_update(W, 0.99)
    for k in ("teacher.encoder.pos_embed", "teacher.encoder.blocks.1.mlp.fc1.weight", "teacher.projector.0.weight"):
        v = W[k].reshape(-1)
        assert rel(v[sample_idx(v.numel())].numpy(), G["ema/" + k]) < 1e-6, k
