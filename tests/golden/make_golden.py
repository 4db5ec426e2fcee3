#!/usr/bin/env python3
"""Generate golden fixtures by importing the upstream reference (read-only at /root/reference).

Runs ONLY in the build container (the reference never travels to the GPU box); the produced ``*.npz`` files are
data (inputs are rebuilt from oracle.atst_oracle.recipe_* by seed; expected outputs are stored) and are committed.

    python tests/golden/make_golden.py            # rewrites tests/golden/*.npz

Shims applied to the reference, in this harness only:
  * torch.distributed gloo group of world size 1 (compute_var all-reduces unconditionally, byol.py:48-50)
  * Tensor.cuda -> identity (hard-coded .cuda() at byol.py:44)
  * stub ``fairseq.data.data_utils`` module (random_mask.py:1 imports it at module top)
  * torch.rand is wrapped to RECORD the DropPath draws so that the keep decisions can be injected elsewhere
"""
import os
import sys
import types
from functools import partial

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
sys.path.insert(0, "/root/reference/audiossl/methods/atstframe")

from oracle import atst_oracle as O  # noqa: E402

torch.set_num_threads(8)
if not torch.distributed.is_initialized():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29512")
    torch.distributed.init_process_group("gloo", rank=0, world_size=1)
torch.Tensor.cuda = lambda self, *a, **k: self

fs = types.ModuleType("fairseq"); fsd = types.ModuleType("fairseq.data"); fsdu = types.ModuleType("fairseq.data.data_utils")
fsdu.compute_mask_indices = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("fairseq stub"))
sys.modules.update({"fairseq": fs, "fairseq.data": fsd, "fairseq.data.data_utils": fsdu})

from audiossl.models.atst.atst import ATST  # noqa: E402
from audiossl.models.atst.audio_transformer import AST  # noqa: E402
from audiossl.modules import transformer as ref_tr  # noqa: E402
from audiossl.utils import common as ref_common  # noqa: E402
torch.set_num_threads(8)   # `import audiossl` exports OMP/MKL=1; restore for generation speed


class RandRecorder:
    """Records every torch.rand draw (DropPath is the only user on this path)."""
    def __init__(self):
        self.draws = []
        self._orig = torch.rand

    def __enter__(self):
        def rec(*a, **k):
            r = self._orig(*a, **k)
            self.draws.append(r.reshape(-1).clone())
            return r
        torch.rand = rec
        return self

    def __exit__(self, *exc):
        torch.rand = self._orig


def keep_from_draws(draws, depth, rates):
    """draws: 2*(depth-1) tensors [S] in call order (block 0 is nn.Identity).  -> [depth,2,S] float {0,1}."""
    S = draws[0].numel()
    keep = torch.ones(depth, 2, S)
    it = iter(draws)
    for i in range(1, depth):
        for j in range(2):
            keep[i, j] = torch.floor((1.0 - rates[i]) + next(it))
    return keep


def sample_idx(n, k=192):
    return np.unique(np.linspace(0, n - 1, num=min(n, k)).astype(np.int64))


def grad_digest(named_params):
    out = {}
    for name, p in named_params:
        if p.grad is None:
            out["gnone/" + name] = np.zeros(0, np.float32)
            continue
        g = p.grad.detach().reshape(-1).double()
        out["gnorm/" + name] = np.array(float(g.norm()))
        out["gsamp/" + name] = g[sample_idx(g.numel())].float().numpy()
    return out


def load_recipe(model, W):
    sd = {k: v for k, v in W.items()}
    missing, unexpected = model.load_state_dict(sd, strict=True), None
    return missing


def save(name, **arrs):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrs.items()})
    print(f"wrote {path}  ({os.path.getsize(path)/1024:.0f} KiB, {len(arrs)} arrays)")


# ----------------------------------------------------------------------------------------------------------------------
def gen_clip(name, ncrops, widths, lengths, B=2, seed_w=0, seed_x=1, seed_dp=7, drop=True):
    torch.manual_seed(123)
    model = ATST("small", ncrops=ncrops)
    model.train()
    W = O.recipe_weights("small", seed=seed_w)
    load_recipe(model, W)
    mels = [O.recipe_mel(B, w, seed=seed_x + i) for i, w in enumerate(widths)]
    lens = [torch.tensor(l, dtype=torch.int64) for l in lengths]
    if not drop:
        for net in (model.student, model.teacher):
            for blk in net.encoder.blocks:
                blk.drop_path = torch.nn.Identity()
    torch.manual_seed(seed_dp)
    with RandRecorder() as rr:
        t_out = model.teacher(mels[:2], lens[:2])
        n_t = len(rr.draws)
        s_out = model.student(mels, lens)
        loss, std_s, std_t = model.loss_fn(s_out, t_out)
    loss.backward()
    rates = O.drop_path_rates(12)
    arrs = dict(B=B, ncrops=ncrops, widths=np.array(widths), lengths=np.array(lengths), seed_w=seed_w, seed_x=seed_x,
                loss=loss.item(), std_s=std_s.item(), std_t=std_t.item(),
                teacher_out=t_out.detach().numpy()[:8], student_out=s_out.detach().numpy()[:8])
    if drop:
        groups = O.group_views(widths)
        # teacher: groups over first 2 views ; student: over all views.  11*2 draws per group pass.
        per = 22
        tg = O.group_views(widths[:2])
        for gi in range(len(tg)):
            arrs[f"keep_t{gi}"] = keep_from_draws(rr.draws[gi * per:(gi + 1) * per], 12, rates).numpy()
        for gi in range(len(groups)):
            arrs[f"keep_s{gi}"] = keep_from_draws(rr.draws[n_t + gi * per:n_t + (gi + 1) * per], 12, rates).numpy()
        assert len(rr.draws) == per * (len(tg) + len(groups)), len(rr.draws)
    arrs.update(grad_digest(model.student.named_parameters()))
    # BN running stats after one train-mode pass
    for k in ("student.projector.1.running_mean", "student.projector.1.running_var", "student.predictor.1.running_var",
              "teacher.projector.1.running_mean", "teacher.projector.1.running_var"):
        arrs["bn/" + k] = model.state_dict()[k][sample_idx(4096)].numpy()
    # EMA (ref: atst.py:29-34) on the un-stepped student
    model.update_teacher(0.99)
    sd = model.state_dict()
    for k in ("teacher.encoder.pos_embed", "teacher.encoder.blocks.3.mlp.fc1.weight", "teacher.projector.0.weight",
              "teacher.projector.1.running_var"):
        v = sd[k].reshape(-1)
        arrs["ema/" + k] = v[sample_idx(v.numel())].numpy()
    save(name, **arrs)


def gen_blocks(name="clip_depth2_blocks"):
    """depth-2 encoder: bit-exact patch indexing + per-block activations (drop_path 0)."""
    enc = AST(depth=2, embed_dim=384, num_heads=6, patch_h=64, patch_w=4, qkv_bias=False,
              norm_layer=partial(torch.nn.LayerNorm, eps=1e-6), drop_path_rate=0.0)
    W = O.recipe_weights("small", depth=2, seed=3)
    enc.load_state_dict({k[len("student.encoder."):]: v for k, v in W.items() if k.startswith("student.encoder.")})
    enc.train()
    S = 3
    mel = O.recipe_mel(S, 1001, seed=5)
    length = torch.tensor([1001, 640, 403])
    # integer-valued probe for the patch gather: value encodes (s, f, t)
    probe = (torch.arange(S)[:, None, None, None] * 100000 + torch.arange(64)[None, None, :, None] * 1001
             + torch.arange(1001)[None, None, None, :]).float()
    patches_probe, _, plen = enc.patch_embed(probe, length)
    x, pos, mel_patches, h, w, plen2 = enc.prepare_tokens(mel, None, length)
    mask = ref_tr.get_attention_mask(x, plen2 + 1)
    b0 = enc.blocks[0](x, plen2 + 1)
    b1 = enc.blocks[1](b0, plen2 + 1)
    cls = enc(mel, length=length)
    save(name, S=S, length=length.numpy(), patch_length=plen.numpy(),
         patches_probe=patches_probe.numpy().astype(np.int32)[:, ::25],      # every 25th patch, all 256 entries
         tokens=x.detach().numpy()[:, ::10, ::4], attn_mask_row=mask[:, 0, 0, :].numpy(),
         block0=b0.detach().numpy()[:, ::10, ::4], block1=b1.detach().numpy()[:, ::10, ::4], cls=cls.detach().numpy())


def gen_frame(name="frame_small"):
    """FrameATST needs pytorch_lightning at import (methods/atstframe/model.py:1) -> assemble it from its parts:
    FrameAST_small + frame MultiCropWrapper + frame ByolLoss, exactly as model.py:24-76 does."""
    import audio_transformer as fat      # methods/atstframe/audio_transformer.py
    import byol as fbyol                 # methods/atstframe/byol.py
    torch.manual_seed(5)
    student = fbyol.MultiCropWrapper(fat.FrameAST_small(pos_type="cut", patch_embed="Linear"), 384, predictor=True)
    teacher = fbyol.MultiCropWrapper(fat.FrameAST_small(pos_type="cut", patch_embed="Linear"), 384, predictor=False)
    loss_fn = fbyol.ByolLoss(symmetric=True)
    W = O.recipe_weights("small", frame=True, seed=11)
    student.load_state_dict({k[len("student."):]: v for k, v in W.items() if k.startswith("student.")})
    teacher.load_state_dict({k[len("teacher."):]: v for k, v in W.items() if k.startswith("teacher.")})
    for p in teacher.parameters():
        p.requires_grad = False
    student.train(); teacher.train()
    B = 2
    mels = [O.recipe_mel(B, 1001, seed=21), O.recipe_mel(B, 1001, seed=22)]
    lens = [torch.tensor([1001, 702]), torch.tensor([1001, 702])]
    rs = np.random.RandomState(99)
    m = torch.from_numpy(np.stack([O.block_mask(250, 0.65, 5, rng=rs) for _ in range(B)]))
    masks = [m, m]
    torch.manual_seed(17)
    with RandRecorder() as rr:
        tea = teacher(mels, lens, masks, False)
        n_t = len(rr.draws)
        stu = student(mels, lens, masks, True)
        loss, std_s, std_t = loss_fn(stu, tea)
    loss.backward()
    rates = O.drop_path_rates(12)
    arrs = dict(B=B, lengths=np.stack([l.numpy() for l in lens]), mask=m.numpy(), M=stu.shape[0],
                loss=loss.item(), std_s=std_s.item(), std_t=std_t.item(),
                teacher_out=tea.detach().numpy()[::7], student_out=stu.detach().numpy()[::7],
                keep_t0=keep_from_draws(rr.draws[:n_t], 12, rates).numpy(),
                keep_s0=keep_from_draws(rr.draws[n_t:], 12, rates).numpy())
    arrs.update(grad_digest(student.named_parameters()))
    save(name, **arrs)


def gen_frame_asym(name="frame_small_asym"):
    """FrameATST(symmetric=False): teacher on view 0 (unmasked), student on view 1 (masked), ByolLoss(symmetric=False)
    (methods/atstframe/model.py:73-76, byol.py:83-84); assembled from its parts like gen_frame."""
    import audio_transformer as fat
    import byol as fbyol
    torch.manual_seed(5)
    student = fbyol.MultiCropWrapper(fat.FrameAST_small(pos_type="cut", patch_embed="Linear"), 384, predictor=True)
    teacher = fbyol.MultiCropWrapper(fat.FrameAST_small(pos_type="cut", patch_embed="Linear"), 384, predictor=False)
    loss_fn = fbyol.ByolLoss(symmetric=False)
    W = O.recipe_weights("small", frame=True, seed=13)
    student.load_state_dict({k[len("student."):]: v for k, v in W.items() if k.startswith("student.")})
    teacher.load_state_dict({k[len("teacher."):]: v for k, v in W.items() if k.startswith("teacher.")})
    for p in teacher.parameters():
        p.requires_grad = False
    student.train(); teacher.train()
    B = 4
    mels = [O.recipe_mel(B, 1001, seed=31), O.recipe_mel(B, 1001, seed=32)]
    lens = [torch.tensor([1001, 702, 1001, 850])] * 2
    rs = np.random.RandomState(77)
    m = torch.from_numpy(np.stack([O.block_mask(250, 0.65, 5, rng=rs) for _ in range(B)]))
    masks = [m, m]
    torch.manual_seed(19)
    with RandRecorder() as rr:
        tea = teacher(mels[:1], lens[:1], masks[:1], False)
        n_t = len(rr.draws)
        stu = student(mels[1:], lens[1:], masks[1:], True)
        loss, std_s, std_t = loss_fn(stu, tea)
    loss.backward()
    rates = O.drop_path_rates(12)
    arrs = dict(B=B, lengths=np.stack([l.numpy() for l in lens]), mask=m.numpy(), M=stu.shape[0],
                loss=loss.item(), std_s=std_s.item(), std_t=std_t.item(),
                teacher_out=tea.detach().numpy()[::7], student_out=stu.detach().numpy()[::7],
                keep_t0=keep_from_draws(rr.draws[:n_t], 12, rates).numpy(),
                keep_s0=keep_from_draws(rr.draws[n_t:], 12, rates).numpy())
    arrs.update(grad_digest(student.named_parameters()))
    save(name, **arrs)


def gen_frame_cnn(name="frame_cnn_patch_embed"):
    """FrameAST_small(patch_embed="CNN") (atstframe/audio_transformer.py:57-74): Conv2d(1, d, (64, 4), stride (64, 4)) patch
    embedding.  Parameter names / shapes of the reference module and its eval-mode frame features on recipe weights whose
    conv kernel is the recipe's Linear weight viewed as [d, 1, 64, 4]."""
    import audio_transformer as fat
    enc = fat.FrameAST_small(pos_type="cut", patch_embed="CNN")
    W = O.recipe_weights("small", frame=True, seed=81)
    sd = {}
    for k, v in W.items():
        if not k.startswith("teacher.encoder."):
            continue
        k = k[len("teacher.encoder."):]
        if k == "patch_embed.patch_embed.weight":
            sd["patch_embed.proj.weight"] = v.reshape(v.shape[0], 1, 64, 4).clone()
        elif k == "patch_embed.patch_embed.bias":
            sd["patch_embed.proj.bias"] = v.clone()
        else:
            sd[k] = v
    enc.load_state_dict(sd)
    enc.eval()
    x, length = O.recipe_mel(2, 1001, seed=83), torch.tensor([1001, 650])
    with torch.no_grad():
        scene = enc.get_intermediate_layers(x, length, 2, scene=True)
        frames = enc.get_intermediate_layers(x, length, 1, scene=False)
    names = np.array([n for n, _ in enc.named_parameters()])
    shapes = np.array([" ".join(str(d) for d in p.shape) for _, p in enc.named_parameters()])
    save(name, length=length.numpy(), scene=scene.numpy(), frames=frames.numpy()[:, ::5, ::4], param_names=names, param_shapes=shapes)


def gen_encoder_grad(name="clip_encoder_grad"):
    """Encoder-only gradient pin that does not pass through the tiny-batch BatchNorm (which is ill-conditioned at B=2,
    see DESIGN.md "Precision"): L = sum(CLS * R) for a fixed R, full-depth AST_small, ragged lengths, DropPath on."""
    from audiossl.models.atst.audio_transformer import AST_small
    torch.manual_seed(3)
    enc = AST_small()
    W = O.recipe_weights("small", seed=21)
    enc.load_state_dict({k[len("student.encoder."):]: v for k, v in W.items() if k.startswith("student.encoder.")})
    enc.train()
    S = 4
    mel = O.recipe_mel(S, 1001, seed=23)
    length = torch.tensor([1001, 777, 1001, 530])
    R = torch.from_numpy(np.random.default_rng(29).standard_normal((S, 384)).astype(np.float32))
    torch.manual_seed(31)
    with RandRecorder() as rr:
        cls = enc(mel, length=length)
    (cls * R).sum().backward()
    arrs = dict(S=S, length=length.numpy(), cls=cls.detach().numpy(),
                keep=keep_from_draws(rr.draws, 12, O.drop_path_rates(12)).numpy())
    arrs.update(grad_digest(enc.named_parameters()))
    save(name, **arrs)


def gen_infer(name="clip_inference_api"):
    """Inference API consumed by the downstream harness (downstream/model.py:34-41), eval mode."""
    from audiossl.models.atst.audio_transformer import AST_small
    enc = AST_small()
    W = O.recipe_weights("small", seed=33)
    enc.load_state_dict({k[len("student.encoder."):]: v for k, v in W.items() if k.startswith("student.encoder.")})
    enc.eval()
    x = O.recipe_mel(2, 1001, seed=35)
    length = torch.tensor([1001, 700])
    with torch.no_grad():
        emb = enc.get_intermediate_layers_chunks(x, length, 2, 601, True)
        emb_cls = enc.get_intermediate_layers_chunks(x, length, 1, 601, False)
        layers = enc.get_intermediate_layers(x, length, 2)
        cls = enc(x, length=length)
    save(name, length=length.numpy(), emb=emb.numpy(), emb_cls=emb_cls.numpy(), cls=cls.numpy(),
         layer_last=layers[-1].numpy()[:, ::10, ::4], layer_prev=layers[0].numpy()[:, ::10, ::4])


def gen_frame_infer(name="frame_inference_api"):
    """FrameAST.get_intermediate_layers (atstframe/audio_transformer.py:259-281) as used by atstframe/embedding.py:75,121:
    scene=True -> masked mean over valid frames of norm_frame(block output), last n blocks concatenated; scene=False -> the
    frame sequences themselves.  Eval mode (no DropPath)."""
    import audio_transformer as fat
    enc = fat.FrameAST_small(pos_type="cut", patch_embed="Linear")
    W = O.recipe_weights("small", frame=True, seed=61)
    enc.load_state_dict({k[len("teacher.encoder."):]: v for k, v in W.items() if k.startswith("teacher.encoder.")})
    enc.eval()
    x = O.recipe_mel(3, 1001, seed=63)
    length = torch.tensor([1001, 640, 88])
    xs = O.recipe_mel(2, 401, seed=65)
    ls = torch.tensor([401, 401])
    with torch.no_grad():
        scene = enc.get_intermediate_layers(x, length, n=3, scene=True)
        frames = enc.get_intermediate_layers(x, length, n=2, scene=False)
        scene_short = enc.get_intermediate_layers(xs, ls, n=12, scene=True)
    save(name, length=length.numpy(), scene=scene.numpy(), frames=frames.numpy()[:, ::5, ::4], frames_shape=np.array(frames.shape),
         scene_short=scene_short.numpy())


def gen_sched(name="schedules"):
    lr = ref_common.cosine_scheduler_step(5e-4 * 4 * 384 / 256, 1e-6, 39100, 1300)
    wd = ref_common.cosine_scheduler_step(0.04, 0.4, 39100, 0)
    ema = ref_common.cosine_scheduler_step(0.99, 1, 39100, 0)
    idx = np.array([0, 1, 2, 649, 1299, 1300, 1301, 5000, 20000, 39098, 39099])
    model = ATST("small")
    reg, noreg = ref_common.get_params_groups(model.student, debug=True)
    names = [n for n, _ in model.student.named_parameters()]
    sd_keys = list(model.state_dict().keys())
    save(name, idx=idx, lr=lr[idx], wd=wd[idx], ema=ema[idx], lr_len=len(lr),
         reg=np.array(reg), noreg=np.array(noreg), param_names=np.array(names), state_dict_keys=np.array(sd_keys),
         n_student=sum(p.numel() for p in model.student.parameters()),
         n_teacher=sum(p.numel() for p in model.teacher.parameters()))


def gen_aug(name="aug_byol_a"):
    """RandomResizeCrop / Mixup of the reference (audiossl/transforms/byol_a.py) with seeded global RNGs; the draws are
    recovered by replaying the same seeds, so that they can be injected into the build's kernels."""
    import random
    from audiossl.transforms import byol_a
    out = {}
    widths = [401, 101, 401, 1001, 101, 401]
    params, outs = [], []
    for k, W in enumerate(widths):
        x = O.recipe_mel(1, W, seed=200 + k)[0]                                    # [1, 64, W]
        rrc = byol_a.RandomResizeCrop(virtual_crop_scale=(1.0, 1.5), freq_scale=(0.6, 1.5), time_scale=(0.6, 1.5))
        CH, CW = int(64 * 1.0), int(W * 1.5)
        np.random.seed(900 + k); random.seed(900 + k)
        params.append(byol_a.RandomResizeCrop.get_params((CH, CW), (64, W), rrc.time_scale, rrc.freq_scale))
        np.random.seed(900 + k); random.seed(900 + k)
        y = rrc(x)
        assert y.shape == x.shape
        out[f"rrc_out{k}"] = y[0, ::2, ::3].numpy()
    out["rrc_widths"] = np.array(widths); out["rrc_params"] = np.array(params, dtype=np.int32)
    # Mixup: equal lengths, input shorter than the bank entry, input longer than the bank entry
    cases = [(401, 401), (101, 401), (401, 101)]
    mix_meta = []
    for k, (Wx, Wz) in enumerate(cases):
        z = O.recipe_mel(1, Wz, seed=300 + k)[0]; x = O.recipe_mel(1, Wx, seed=310 + k)[0]
        m = byol_a.Mixup(ratio=0.4, n_memory=2000)
        np.random.seed(950 + k)
        assert torch.equal(m(z), z)                                                # empty bank: identity, z enters the bank
        y = m(x)
        np.random.seed(950 + k)                                                    # replay the draws
        np.random.random()                                                         # alpha of the first call
        a = 0.4 * np.random.random(); idx = np.random.randint(1)
        start = np.random.randint(0, abs(Wz - Wx)) if Wx != Wz else 0
        mix_meta.append((Wx, Wz, idx, start))
        out[f"mix_alpha{k}"] = np.float64(a)
        out[f"mix_out{k}"] = y[0, ::2, ::3].numpy()
    out["mix_meta"] = np.array(mix_meta, dtype=np.int32)
    save(name, **out)


def gen_base_encoder_grad(name="base_depth3_encoder_grad"):
    """ATST-base geometry pinned to the reference: the encoder class AST_base builds (audio_transformer.py:371-374:
    embed_dim 768, 12 heads, 64 x 4 patches, qkv_bias False, LayerNorm eps 1e-6) at depth 3, ragged lengths, DropPath 0.1 recorded;
    per-block activations (sampled), CLS, and the gradient of the smooth objective L = sum(CLS * R)."""
    depth, S = 3, 6
    torch.manual_seed(3)
    enc = AST(depth=depth, embed_dim=768, num_heads=12, patch_h=64, patch_w=4, qkv_bias=False,
              norm_layer=partial(torch.nn.LayerNorm, eps=1e-6), drop_path_rate=0.1)
    W = O.recipe_weights("base", depth=depth, seed=31)
    enc.load_state_dict({k[len("student.encoder."):]: v for k, v in W.items() if k.startswith("student.encoder.")})
    enc.train()
    mel = O.recipe_mel(S, 1001, seed=33)
    length = torch.tensor([1001, 1001, 777, 640, 1001, 405])
    R = torch.from_numpy(np.random.default_rng(35).standard_normal((S, 768)).astype(np.float32))
    rates = [x.item() for x in torch.linspace(0, 0.1, depth)]
    torch.manual_seed(37)
    with RandRecorder() as rr:
        x, pos, mel_patches, h, w, plen = enc.prepare_tokens(mel, None, length)
        blocks = []
        for blk in enc.blocks:
            x = blk(x, plen + 1)
            blocks.append(x)
        cls = enc.norm(x)[:, 0]
    (cls * R).sum().backward()
    keep = keep_from_draws(rr.draws, depth, rates)
    # the same forward through AST.forward with the same draws replayed must give the same CLS (the block loop above IS forward())
    torch.manual_seed(37)
    with torch.no_grad():
        cls2 = enc(mel, length=length)
    assert torch.allclose(cls2, cls.detach(), atol=1e-6), float((cls2 - cls).abs().max())
    arrs = dict(S=S, depth=depth, length=length.numpy(), patch_length=plen.numpy(), cls=cls.detach().numpy(), keep=keep.numpy(), rates=np.array(rates))
    for i, b in enumerate(blocks):
        arrs[f"block{i}"] = b.detach().numpy()[:, ::10, ::8]
    arrs.update(grad_digest(enc.named_parameters()))
    save(name, **arrs)


def gen_base_step(name="base_2views_depth2"):
    """One 2-view training step of an ATST("base")-shaped model at depth 2: the reference's MultiCropWrapper (byol.py:75-121) around
    AST(embed_dim 768, 12 heads) -- what ATST.__init__ builds for arch == "base" (atst.py:11-17) with a shallower encoder -- its
    ByolLoss(2), backward, BN running statistics and the EMA update (atst.py:29-34)."""
    from audiossl.models.atst.byol import MultiCropWrapper, ByolLoss
    depth, B = 2, 16
    mk = lambda: AST(depth=depth, embed_dim=768, num_heads=12, patch_h=64, patch_w=4, qkv_bias=False,
                     norm_layer=partial(torch.nn.LayerNorm, eps=1e-6), drop_path_rate=0.1)
    torch.manual_seed(123)
    student = MultiCropWrapper(mk(), 768, predictor=True)
    teacher = MultiCropWrapper(mk(), 768, predictor=False)
    for p in teacher.parameters():
        p.requires_grad = False
    loss_fn = ByolLoss(2)
    W = O.recipe_weights("base", depth=depth, seed=41)
    student.load_state_dict({k[len("student."):]: v for k, v in W.items() if k.startswith("student.")})
    teacher.load_state_dict({k[len("teacher."):]: v for k, v in W.items() if k.startswith("teacher.")})
    student.train(); teacher.train()
    widths = [1001, 1001]
    mels = [O.recipe_mel(B, w, seed=43 + i) for i, w in enumerate(widths)]
    lens = [torch.tensor([1001] * B), torch.tensor([1001 - 60 * (i % 4) for i in range(B)])]
    rates = [x.item() for x in torch.linspace(0, 0.1, depth)]
    torch.manual_seed(47)
    with RandRecorder() as rr:
        t_out = teacher(mels, lens)
        n_t = len(rr.draws)
        s_out = student(mels, lens)
        loss, std_s, std_t = loss_fn(s_out, t_out)
    loss.backward()
    arrs = dict(B=B, depth=depth, widths=np.array(widths), lengths=np.stack([l.numpy() for l in lens]), loss=loss.item(), std_s=std_s.item(),
                std_t=std_t.item(), teacher_out=t_out.detach().numpy()[:8], student_out=s_out.detach().numpy()[:8],
                keep_t0=keep_from_draws(rr.draws[:n_t], depth, rates).numpy(), keep_s0=keep_from_draws(rr.draws[n_t:], depth, rates).numpy())
    arrs.update(grad_digest(student.named_parameters()))
    sd = {**{"student." + k: v for k, v in student.state_dict().items()}, **{"teacher." + k: v for k, v in teacher.state_dict().items()}}
    for k in ("student.projector.1.running_mean", "student.projector.1.running_var", "student.predictor.1.running_var",
              "teacher.projector.1.running_mean", "teacher.projector.1.running_var"):
        arrs["bn/" + k] = sd[k][sample_idx(4096)].numpy()
    with torch.no_grad():                                              # ATST.update_teacher (atst.py:29-34) on the un-stepped student
        for net in ("encoder", "projector"):
            for pq, pk in zip(getattr(student, net).parameters(), getattr(teacher, net).parameters()):
                pk.data.mul_(0.99).add_((1 - 0.99) * pq.detach().data)
    tsd = teacher.state_dict()
    for k in ("encoder.pos_embed", "encoder.blocks.1.mlp.fc1.weight", "projector.0.weight"):
        v = tsd[k].reshape(-1)
        arrs["ema/teacher." + k] = v[sample_idx(v.numel())].numpy()
    save(name, **arrs)


def gen_frame_base(name="frame_base_depth2"):
    """ATST-Frame *base* (methods/atstframe/train_base.sh:4-17, audio_transformer.py:287-288: FrameAST(embed_dim 768, 12 heads)) at depth 2: the
    reference's frame MultiCropWrapper + ByolLoss(symmetric) around it (model.py:24-76), B = 8 sequences per view with ragged lengths, one block mask per
    sequence shared by both views -- ~2.6 k masked head rows through the 768 -> 4096 -> 256 heads.  Assembled from its parts like gen_frame."""
    import audio_transformer as fat      # methods/atstframe/audio_transformer.py
    import byol as fbyol                 # methods/atstframe/byol.py
    depth, B = 2, 8
    mk = lambda: fat.FrameAST(patch_h=64, patch_w=4, embed_dim=768, depth=depth, num_heads=12, qkv_bias=False,
                              norm_layer=partial(torch.nn.LayerNorm, eps=1e-6), pos_type="cut", patch_embed="Linear")
    torch.manual_seed(5)
    student = fbyol.MultiCropWrapper(mk(), 768, predictor=True)
    teacher = fbyol.MultiCropWrapper(mk(), 768, predictor=False)
    loss_fn = fbyol.ByolLoss(symmetric=True)
    W = O.recipe_weights("base", depth=depth, frame=True, seed=71)
    student.load_state_dict({k[len("student."):]: v for k, v in W.items() if k.startswith("student.")})
    teacher.load_state_dict({k[len("teacher."):]: v for k, v in W.items() if k.startswith("teacher.")})
    for p in teacher.parameters():
        p.requires_grad = False
    student.train(); teacher.train()
    mels = [O.recipe_mel(B, 1001, seed=73), O.recipe_mel(B, 1001, seed=74)]
    ln = torch.tensor([1001, 702, 1001, 941, 1001, 523, 1001, 881])
    lens = [ln, ln.clone()]
    rs = np.random.RandomState(77)
    m = torch.from_numpy(np.stack([O.block_mask(250, 0.65, 5, rng=rs) for _ in range(B)]))
    masks = [m, m]
    rates = [x.item() for x in torch.linspace(0, 0.1, depth)]
    torch.manual_seed(79)
    with RandRecorder() as rr:
        tea = teacher(mels, lens, masks, False)
        n_t = len(rr.draws)
        stu = student(mels, lens, masks, True)
        loss, std_s, std_t = loss_fn(stu, tea)
    loss.backward()
    arrs = dict(B=B, depth=depth, lengths=np.stack([l.numpy() for l in lens]), mask=m.numpy(), M=stu.shape[0],
                loss=loss.item(), std_s=std_s.item(), std_t=std_t.item(),
                teacher_out=tea.detach().numpy()[::23], student_out=stu.detach().numpy()[::23],
                keep_t0=keep_from_draws(rr.draws[:n_t], depth, rates).numpy(),
                keep_s0=keep_from_draws(rr.draws[n_t:], depth, rates).numpy())
    arrs.update(grad_digest(student.named_parameters()))
    save(name, **arrs)


if __name__ == "__main__":
    which = sys.argv[1:] or ["blocks", "clip2", "clip2_nodrop", "clip6", "frame", "sched", "encgrad", "clip2_b16", "clip2_b64", "infer", "aug", "frame_infer", "frame_asym", "frame_cnn", "base_encgrad", "base_step", "frame_base"]
    if "aug" in which:
        gen_aug()
    if "frame_infer" in which:
        gen_frame_infer()
    if "blocks" in which:
        gen_blocks()
    if "clip2" in which:
        gen_clip("clip_small_2views", 2, [1001, 1001], [[1001, 801], [901, 1001]])
    if "clip2_nodrop" in which:
        gen_clip("clip_small_2views_nodrop", 2, [1001, 1001], [[1001, 1001], [1001, 1001]], drop=False, seed_x=31)
    if "clip6" in which:
        gen_clip("clip_small_6crops", 6, [1001, 1001, 101, 101, 101, 101],
                 [[1001, 1001], [1001, 1001], [101, 101], [101, 77], [101, 101], [101, 101]], seed_x=41)
    if "encgrad" in which:
        gen_encoder_grad()
    if "infer" in which:
        gen_infer()
    if "clip2_b16" in which:
        L = [1001] * 16
        L2 = [1001 - 40 * (i % 5) for i in range(16)]
        gen_clip("clip_small_2views_b16", 2, [1001, 1001], [L, L2], B=16, seed_x=51)
    if "clip2_b64" in which:
        L = [1001] * 64
        L2 = [1001 - 40 * (i % 5) for i in range(64)]
        gen_clip("clip_small_2views_b64", 2, [1001, 1001], [L, L2], B=64, seed_x=71)
    if "frame" in which:
        gen_frame()
    if "frame_asym" in which:
        gen_frame_asym()
    if "frame_cnn" in which:
        gen_frame_cnn()
    if "sched" in which:
        gen_sched()
    if "base_encgrad" in which:
        gen_base_encoder_grad()
    if "base_step" in which:
        gen_base_step()
    if "frame_base" in which:
        gen_frame_base()
