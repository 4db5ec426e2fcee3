"""Reference-shaped host API on the GPU: module tree / state_dict keys, autograd glue, fused optimizer through the
Lightning-style hook order, Lightning-format checkpoint round trip, inference API vs the reference golden, batched
augmentation stage."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from audiossl_amd.methods.atst.data import ATSTDataModule  # noqa: E402
from audiossl_amd.methods.atst.model import ATSTLightningModule  # noqa: E402
from audiossl_amd.methods.atst.transform import ATSTBatchViews, ATSTTrainTransform  # noqa: E402
from audiossl_amd.methods.atstframe.model import FrameATSTLightningModule  # noqa: E402
from audiossl_amd.models.atst import ATST  # noqa: E402
from audiossl_amd.trainer import Trainer, load_checkpoint, save_checkpoint  # noqa: E402
from oracle import atst_oracle as O  # noqa: E402

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


def test_state_dict_keys_match_reference_and_load():
    G = np.load(os.path.join(GOLD, "schedules.npz"))
    model = ATST("small")
    assert list(model.state_dict().keys()) == list(G["state_dict_keys"])
    W = O.recipe_weights("small", seed=33)
    model.load_state_dict(W)                                             # reference-keyed state_dict loads strictly
    got = model.state_dict()
    for k in ("student.encoder.blocks.5.attn.qkv.weight", "teacher.projector.1.running_var", "student.predictor.3.weight"):
        assert torch.equal(got[k].cpu(), W[k])
    with pytest.raises(RuntimeError):
        ATST("huge")                                                     # ref: atst.py:17


def test_inference_api_vs_reference_golden():
    G = np.load(os.path.join(GOLD, "clip_inference_api.npz"))
    model = ATST("small")
    W = O.recipe_weights("small", seed=33)
    model.load_state_dict(W)
    enc = model.student.encoder
    x, length = O.recipe_mel(2, 1001, seed=35), torch.from_numpy(G["length"])
    assert enc.embed_dim == 384
    assert rel(enc(x, length=length).cpu().numpy(), G["cls"]) < 1e-2
    layers = enc.get_intermediate_layers(x, length, 2)
    assert layers[-1].shape == (2, 251, 384)
    assert rel(layers[-1].cpu().numpy()[:, ::10, ::4], G["layer_last"]) < 1.0e-2
    assert rel(layers[0].cpu().numpy()[:, ::10, ::4], G["layer_prev"]) < 1.0e-2
    emb = enc.get_intermediate_layers_chunks(x, length, 2, 601, True)
    assert emb.shape == (2, 2 * 2 * 384) and rel(emb.cpu().numpy(), G["emb"]) < 1.0e-2
    assert rel(enc.get_intermediate_layers_chunks(x, length, 1, 601, False).cpu().numpy(), G["emb_cls"]) < 1.0e-2


def test_autograd_glue_and_lightning_hook_order(tmp_path):
    torch.manual_seed(0)
    module = ATSTLightningModule(arch="small", learning_rate=1e-3, warmup_steps=2, max_steps=6, ema=0.99, nproc=1, save_path="x")
    assert module.hparams["nproc"] == 1 and module.hparams["max_steps"] == 6             # unknown kwargs are swallowed
    B = 4
    views = ATSTBatchViews()

    class DS(torch.utils.data.Dataset):
        def __init__(self):
            self.t = ATSTTrainTransform(anchor_len=(2.0, 2.0), positive_len=(2.0, 2.0))

        def __len__(self):
            return 3 * B

        def __getitem__(self, i):
            g = torch.Generator().manual_seed(i)
            return self.t(0.1 * torch.randn(1, 48000, generator=g)), torch.zeros(1)
    loader = torch.utils.data.DataLoader(DS(), batch_size=B, drop_last=True)
    (crops, lengths), _ = next(iter(loader))
    assert crops[0].shape == (B, 1, 32000) and int(lengths[0][0]) == 201                 # 2 s -> 201 frames
    before = module.model.student.encoder.blocks[3].mlp.fc1.weight.detach().clone()
    t_before = module.model.teacher.encoder.pos_embed.detach().clone()
    trainer = Trainer(max_steps=3, default_root_dir=str(tmp_path), every_n_epochs=1, log_every_n_steps=1, batch_hook=views)
    trainer.fit(module, train_dataloaders=loader)
    assert module.global_step == 3 and module.model.engine.opt_step == 3
    p = module.model.student.encoder.blocks[3].mlp.fc1.weight
    assert p.grad is not None and p.grad.shape == p.shape and torch.isfinite(p.grad).all()
    assert module.model.student.encoder.mask_embed.grad is None                          # reference: grad is None
    assert not torch.equal(p.detach(), before) and not torch.equal(module.model.teacher.encoder.pos_embed, t_before)
    assert {"loss", "std_cls_s", "std_cls_t", "ema", "step", "wd", "lr"} <= set(module.logged)
    assert abs(module.logged["lr"] - module.mylr_scheduler[2]) < 1e-12                   # schedule index = pre-increment step
    # Lightning-format checkpoint round trip
    ck = os.path.join(str(tmp_path), "last.ckpt")
    assert os.path.exists(ck)
    raw = torch.load(ck, map_location="cpu", weights_only=False)
    assert "pytorch-lightning_version" in raw and raw["global_step"] == 3 and "train_len" not in raw["hyper_parameters"]
    assert all(k.startswith("model.") for k in raw["state_dict"])
    again = ATSTLightningModule.load_from_checkpoint(ck)
    for (k, a), (_, b) in zip(module.state_dict().items(), again.state_dict().items()):
        assert torch.equal(a.cpu(), b.cpu()), k
    assert again.global_step == 3 and again.model.engine.opt_step == 3


def test_external_optimizer_compat_path():
    """Any torch optimizer can step the parameter views; shadows are refreshed from the version counters."""
    model = ATST("small", depth=2)
    opt = torch.optim.SGD([p for p in model.student.parameters() if p.requires_grad], lr=0.1)
    mels = [O.recipe_mel(4, 1001, seed=1).cuda(), O.recipe_mel(4, 1001, seed=2).cuda()]
    lens = [torch.full((4,), 1001)] * 2
    l0 = model(mels, lens)[0]
    l0.backward()
    opt.step()
    model.update_teacher(0.9)
    l1 = model(mels, lens)[0]
    assert torch.isfinite(l1) and abs(l1.item() - l0.item()) > 1e-6                      # the step took effect in the HIP path


def test_frame_module_step():
    module = FrameATSTLightningModule(arch="small", max_steps=4, warmup_steps=1)
    module.trainer = type("T", (), {"optimizers": module.configure_optimizers()})()
    B = 2
    mels = [O.recipe_mel(B, 1001, seed=3).cuda()] * 2
    rs = np.random.RandomState(0)
    m = torch.from_numpy(np.stack([O.block_mask(250, 0.65, 5, rng=rs) for _ in range(B)]))
    loss = module.training_step(((mels, [torch.full((B,), 1001)] * 2, [m, m]), None), 0)
    loss.backward()
    assert module.model.student.encoder.mask_embed.grad is not None                      # trained in ATST-Frame
    module.trainer.optimizers[0].step()
    assert torch.isfinite(loss)


def test_frame_inference_api_vs_reference_golden(tmp_path):
    """FrameAST.get_intermediate_layers semantics (scene / frame outputs) against the reference golden, then the
    embedding helpers of atstframe/embedding.py on a saved checkpoint."""
    from audiossl_amd.models.atst import FrameATST
    G = np.load(os.path.join(GOLD, "frame_inference_api.npz"))
    model = FrameATST("small")
    model.load_state_dict(O.recipe_weights("small", frame=True, seed=61))
    enc = model.teacher.encoder
    x, length = O.recipe_mel(3, 1001, seed=63), torch.from_numpy(G["length"])
    scene = enc.get_intermediate_layers(x, length, n=3, scene=True)
    assert scene.shape == (3, 3 * 384) and rel(scene.cpu().numpy(), G["scene"]) < 1.0e-2
    frames = enc.get_intermediate_layers(x, length, n=2, scene=False)
    assert tuple(frames.shape) == tuple(G["frames_shape"])
    assert rel(frames.cpu().numpy()[:, ::5, ::4], G["frames"]) < 1.0e-2
    short = enc.get_intermediate_layers(O.recipe_mel(2, 401, seed=65), torch.tensor([401, 401]), n=12)
    assert rel(short.cpu().numpy(), G["scene_short"]) < 1.0e-2
    # embedding helpers: 13 s of audio -> two chunks (1001 + 300 frames)
    from audiossl_amd.methods.atstframe import embedding as E
    from audiossl_amd.methods.atstframe.model import FrameATSTLightningModule
    from audiossl_amd.trainer import save_checkpoint
    module = FrameATSTLightningModule(arch="small", max_steps=10, warmup_steps=2)
    module.model.load_state_dict(O.recipe_weights("small", frame=True, seed=61))
    path = str(tmp_path / "frame.ckpt")
    save_checkpoint(path, module)
    enc2 = E.load_model(path)
    wave = O.recipe_wave(2, 13 * 16000, seed=7).cuda()
    emb = E.get_scene_embedding(wave, enc2)
    assert emb.shape == (2, 12 * 384) and torch.isfinite(emb).all()
    ts_emb, ts = E.get_timestamp_embedding(wave, enc2)
    assert ts_emb.shape == (2, 250 + 75, 12 * 384) and ts.shape == (2, 325) and float(ts[0, 1]) == 40.0
    mel = enc2.transform(wave[:, :160000 + 160])                   # first chunk alone == first 250 frames of the stream
    first = enc2.get_intermediate_layers(enc2.transform(wave)[..., :1001], torch.tensor([1001, 1001]), n=12, scene=False)
    assert rel(ts_emb[:, :250].cpu().numpy(), first.cpu().numpy()) < 1e-6 and mel.shape[-1] == 1002


def test_batched_augmentations():
    """HIP augmentation kernels, through the host classes with the reference's draws injected, against goldens of
    audiossl/transforms/byol_a.py (tests/golden/make_golden.py gen_aug); then the classes' own sampling."""
    import numpy as np
    from audiossl_amd.transforms import BatchMixup, BatchRandomResizeCrop
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "aug_byol_a.npz"))
    rrc = BatchRandomResizeCrop((1.0, 1.5), rng=np.random.RandomState(0))
    for k, W in enumerate(G["rrc_widths"]):
        x = O.recipe_mel(1, int(W), seed=200 + k).cuda()                               # [1, 1, 64, W]
        y = rrc.apply(x, G["rrc_params"][k][None])
        assert y.shape == x.shape
        assert np.abs(y[0, 0, ::2, ::3].cpu().numpy() - G[f"rrc_out{k}"]).max() < 2e-5
    for k, (Wx, Wz, idx, start) in enumerate(G["mix_meta"]):
        z = O.recipe_mel(1, int(Wz), seed=300 + k).cuda(); x = O.recipe_mel(1, int(Wx), seed=310 + k).cuda()
        mix = BatchMixup(n_memory=8, rng=np.random.RandomState(0))
        assert torch.equal(mix(z), z) and mix.filled == 1                              # empty bank: identity (byol_a.py:98-107)
        zs, xs = (int(start), 0) if Wx < Wz else (0, int(start))
        y = mix.apply(x, [float(G[f"mix_alpha{k}"])], [int(idx)], [zs], [xs])
        assert np.abs(y[0, 0, ::2, ::3].cpu().numpy() - G[f"mix_out{k}"]).max() < 2e-5
    # batched use with the classes' own draws
    x = O.recipe_mel(6, 401, seed=9).cuda()
    mix = BatchMixup(rng=np.random.RandomState(1))
    assert torch.equal(mix(x), x)
    y = mix(x)
    assert y.shape == x.shape and not torch.equal(y, x) and mix.filled == 12 and torch.isfinite(y).all()
    z = BatchRandomResizeCrop((1, 1.5), rng=np.random.RandomState(2))(x)
    assert z.shape == x.shape and torch.isfinite(z).all()
    ident = BatchRandomResizeCrop((1, 1.0), freq_scale=(1.0, 1.0), time_scale=(1.0, 1.0))(x)
    assert rel(ident.cpu().numpy(), x.cpu().numpy()) < 1e-6                              # full-size crop == identity


def test_frame_trainer_end_to_end(tmp_path):
    """FrameATSTDataModule -> Trainer (batch carries (crops, lengths, masks), ref: atstframe/model.py:120) -> GPU stage of the
    transform (mel win 640, clean view + Mixup / frequency-only RandomResizeCrop view, ONE shared mask) -> HIP step,
    checkpoint, resume with the epoch counter restored."""
    from audiossl_amd.methods.atstframe.data import FrameATSTDataModule
    from audiossl_amd.methods.atstframe.model import FrameATSTLightningModule
    from audiossl_amd.trainer import Trainer
    torch.manual_seed(0)
    module = FrameATSTLightningModule(arch="small", learning_rate=1e-3, warmup_steps=2, max_steps=4, ema=0.997)
    dm = FrameATSTDataModule(batch_size_per_gpu=2, num_workers=0, subset=4, win_length=640, aug_tea=False, aug_stu=True,
                             mask_ratio=0.65, mask_type="block", anchor_len=10, mask_len=5)
    hook = dm.transform.batch_views()
    (crops, lengths, masks), _ = next(iter(dm.train_dataloader()))
    views = hook([crops[0].cuda()], lengths)
    assert len(views) == 2 and views[0].shape == (2, 1, 64, 1001) and views[1].shape == (2, 1, 64, 1001)
    again = hook([crops[0].cuda()], lengths)
    assert torch.equal(views[0], again[0])                                     # aug_tea=False: view 0 is the clean mel
    assert not torch.equal(again[0], again[1])                                 # view 1: Mixup (bank now filled) + freq RRC
    trainer = Trainer(max_steps=4, default_root_dir=str(tmp_path), every_n_epochs=1, log_every_n_steps=1, batch_hook=hook)
    trainer.fit(module, datamodule=dm)
    assert module.global_step == 4 and {"loss", "loss_frm", "std_frm_tea", "std_frm_stu"} <= set(module.logged)
    assert module.model.student.encoder.mask_embed.grad is not None and torch.isfinite(module.logged["loss"])
    raw = torch.load(os.path.join(str(tmp_path), "last.ckpt"), map_location="cpu", weights_only=False)
    assert raw["epoch"] == 1 and raw["global_step"] == 4                       # two epochs of two batches; zero-based index of the last
    resumed = FrameATSTLightningModule(arch="small", learning_rate=1e-3, warmup_steps=2, max_steps=6, ema=0.997)
    t2 = Trainer(max_steps=6, default_root_dir=str(tmp_path), every_n_epochs=1, log_every_n_steps=1, batch_hook=hook)
    t2.fit(resumed, datamodule=dm, ckpt_path=os.path.join(str(tmp_path), "last.ckpt"))
    assert resumed.global_step == 6 and resumed.model.engine.opt_step == 6
    assert os.path.exists(os.path.join(str(tmp_path), "checkpoint-epoch=00002.ckpt"))     # epoch numbering continued, nothing overwritten


def test_inference_workspaces_are_bounded():
    """Variable-width evaluation re-uses a small LRU of forward-only workspaces (no activation tape)."""
    model = ATST("small", depth=2)
    enc = model.teacher.encoder
    for w in (401, 405, 409, 413, 417, 421, 425):
        y = enc(O.recipe_mel(2, w, seed=w), length=torch.tensor([w, w - 100]))
        assert y.shape == (2, 384) and torch.isfinite(y).all()
    eng = model.engine
    assert len(eng._infer_passes) <= 4 and not any(k[3] for k in eng._passes)     # no training-sized pass was created
    assert all(not p.train for p in eng._infer_passes.values())


def test_frame_cnn_patch_embed_vs_reference_golden():
    """FrameATST(patch_embed="CNN"): the Conv2d(1, d, (64, 4), stride (64, 4)) patch embedding of the reference
    (atstframe/audio_transformer.py:57-74) is the same contraction as the Linear one; state_dict keys / shapes are the conv's."""
    from audiossl_amd.models.atst import FrameATST
    G = np.load(os.path.join(GOLD, "frame_cnn_patch_embed.npz"))
    model = FrameATST("small", patch_embed="CNN")
    sd = model.state_dict()
    assert sd["teacher.encoder.patch_embed.proj.weight"].shape == (384, 1, 64, 4) and "teacher.encoder.patch_embed.patch_embed.weight" not in sd
    W = O.recipe_weights("small", frame=True, seed=81)
    W2 = {}
    for k, v in W.items():
        if k.endswith("patch_embed.patch_embed.weight"):
            W2[k.replace("patch_embed.patch_embed.weight", "patch_embed.proj.weight")] = v.reshape(v.shape[0], 1, 64, 4)
        elif k.endswith("patch_embed.patch_embed.bias"):
            W2[k.replace("patch_embed.patch_embed.bias", "patch_embed.proj.bias")] = v
        else:
            W2[k] = v
    model.load_state_dict(W2)
    enc = model.teacher.encoder
    x, length = O.recipe_mel(2, 1001, seed=83), torch.from_numpy(G["length"])
    scene = enc.get_intermediate_layers(x, length, n=2, scene=True)
    frames = enc.get_intermediate_layers(x, length, n=1, scene=False)
    assert rel(scene.cpu().numpy(), G["scene"]) < 1.0e-2 and rel(frames.cpu().numpy()[:, ::5, ::4], G["frames"]) < 1.0e-2
    with pytest.raises(NotImplementedError):
        FrameATST("small", patch_embed="MLP")


def test_downstream_harness_attribute_walk_through_aliased_imports(tmp_path):
    """What the unchanged downstream harness does with a pre-training checkpoint (audiossl/methods/atst/downstream/
    train_freeze.py:23-35, downstream/model.py:18-41), written with the reference's OWN import lines after
    install_as_audiossl(): load_from_checkpoint -> .model.teacher.encoder -> .hyper_param / .embed_dim ->
    get_intermediate_layers_chunks; and the legacy branch's AST_small() + load_state_dict."""
    import audiossl_amd
    audiossl_amd.install_as_audiossl()
    try:
        from audiossl.methods.atst.model import ATSTLightningModule as UpstreamName
        from audiossl.models.atst import audio_transformer
        assert UpstreamName is ATSTLightningModule
        pl = ATSTLightningModule(arch="small", max_steps=10, warmup_steps=2, train_len=6.0)
        path = os.path.join(str(tmp_path), "last.ckpt")
        save_checkpoint(path, pl, None, epoch=0)
        s = torch.load(path, weights_only=False)
        assert "pytorch-lightning_version" in s
        pretrained_model = UpstreamName.load_from_checkpoint(path)
        enc = pretrained_model.model.teacher.encoder
        assert isinstance(enc, audio_transformer.AST)
        enc.hyper_param = s["hyper_parameters"]
        assert enc.hyper_param["train_len"] == 6.0 and enc.embed_dim == 384
        chunk_len, n_blocks = int((6.0 * 16000) / 160 + 1), 2                       # downstream/model.py:26
        x = O.recipe_mel(3, 1001, seed=5).cuda()
        length = torch.tensor([1001, 700, 333])
        emb = enc.get_intermediate_layers_chunks(x, length, n_blocks, chunk_len, avgpool=True)
        assert emb.shape == (3, enc.embed_dim * 2 * n_blocks) and torch.isfinite(emb).all()
        # same weights through the legacy branch: a stand-alone AST_small() that takes the teacher encoder's state_dict
        legacy = audio_transformer.AST_small()
        legacy.load_state_dict(enc.state_dict())
        legacy._owner[0].engine.sync_shadows(force=True)
        emb2 = legacy.get_intermediate_layers_chunks(x, length, n_blocks, chunk_len, avgpool=True)
        assert float((emb - emb2).abs().max()) == 0.0
    finally:
        audiossl_amd.compat.uninstall()


def test_frame_transform_and_module_at_the_hires_geometry():
    """The reference threads sr / n_mels / patch_h / patch_w through the Frame transform, the CLI and the module (atstframe/transform.py:14-17,
    train.py:15,50-51).  BASELINE.json configs[4]'s geometry -- 32 kHz, 128 bands, 128 x 8 patches, 10 s = anchor_len 20 in the transform's
    16000-sample units -> 2001 frames -> 250 tokens -- through FrameATSTTrainTransform (mel vs the oracle) and one FrameATSTLightningModule step."""
    from audiossl_amd.methods.atstframe.transform import FrameATSTTrainTransform
    t = FrameATSTTrainTransform(sr=32000, n_mels=128, patch_h=128, patch_w=8, win_length=1024, aug_tea=False, aug_stu=False, mask_type="block",
                                mask_ratio=0.65, anchor_len=20, mask_len=5)
    wave = O.recipe_wave(1, 320000, seed=77)
    views, lengths, masks = t(wave.cuda())
    assert [tuple(v.shape) for v in views] == [(1, 128, 2001)] * 2 and lengths == [2001, 2001] and masks[0].shape == (250,)
    want = O.log_mel(wave, 1024, n_mels=128, sample_rate=32000)[0]
    assert float((views[0].cpu() - want).abs().max()) < 2e-3
    module = FrameATSTLightningModule(arch="small", max_steps=4, warmup_steps=1, spec_h=128, patch_h=128, patch_w=8, spec_w=2001)
    assert module.model.engine.patch_h == 128 and module.model.engine.patch_w == 8
    module.trainer = type("T", (), {"optimizers": module.configure_optimizers()})()
    B = 2
    mels = [O.recipe_mel(B, 2001, seed=3, n_mels=128).cuda()] * 2
    rs = np.random.RandomState(0)
    m = torch.from_numpy(np.stack([O.block_mask(250, 0.65, 5, rng=rs) for _ in range(B)]))
    loss = module.training_step(((mels, [torch.full((B,), 2001)] * 2, [m, m]), None), 0)
    loss.backward()
    module.trainer.optimizers[0].step()
    assert torch.isfinite(loss)
