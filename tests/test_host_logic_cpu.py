"""Host-side logic that needs no GPU: flat parameter layout vs the reference's state_dict, AdamW group flags, schedule
tables, view grouping, token padding, and the product's mask sampler against the oracle's restatement."""
import os

import numpy as np
import pytest
import torch

from audiossl_amd import engine as E
from audiossl_amd.methods.atstframe import random_mask as RM
from audiossl_amd.utils import common as UC
from oracle import atst_oracle as O

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_flat_layout_matches_reference_state_dict():
    G = np.load(os.path.join(GOLD, "schedules.npz"))
    L = E.FlatLayout("small", None, False)
    names = list(L.entries)
    assert names == list(G["param_names"])                        # reference student.named_parameters() order
    assert sum(L.numel(n) for n in names) == int(G["n_student"]) == 26211328
    assert sum(L.numel(n) for n in names if not n.startswith("predictor.")) == int(G["n_teacher"])
    offs = [L.entries[n][0] for n in names]
    assert all(o % E.ALIGN == 0 for o in offs) and offs == sorted(offs)
    for a, b in zip(names[:-1], names[1:]):
        assert L.entries[a][0] + L.numel(a) <= L.entries[b][0]   # no overlap
    assert L.n_teacher == L.entries["predictor.0.weight"][0] and L.n_student % E.ALIGN == 0
    Lf = E.FlatLayout("small", None, True)
    assert "encoder.cls_token" not in Lf.entries and "encoder.norm_frame.weight" in Lf.entries


def test_weight_decay_grouping_matches_reference():
    G = np.load(os.path.join(GOLD, "schedules.npz"))
    L = E.FlatLayout("small", None, False)
    reg = [n for n, (o, s) in L.entries.items() if not (n.endswith(".bias") or len(s) == 1)]
    assert reg == list(G["reg"])
    assert [n for n in L.entries if n not in reg] == list(G["noreg"])

    class Fake(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.zeros(3, 3)); self.bias = torch.nn.Parameter(torch.zeros(3))
            self.g = torch.nn.Parameter(torch.zeros(3)); self.frozen = torch.nn.Parameter(torch.zeros(2, 2), requires_grad=False)
    groups = UC.get_params_groups(Fake())
    assert len(groups[0]["params"]) == 1 and len(groups[1]["params"]) == 2 and groups[1]["weight_decay"] == 0.0
    assert UC.get_params_groups(Fake(), debug=True) == (["w"], ["bias", "g"])


def test_schedules_match_reference():
    G = np.load(os.path.join(GOLD, "schedules.npz"))
    idx = G["idx"]
    assert np.array_equal(UC.cosine_scheduler_step(5e-4 * 4 * 384 / 256, 1e-6, 39100, 1300)[idx], G["lr"])
    assert np.array_equal(UC.cosine_scheduler_step(0.04, 0.4, 39100, 0)[idx], G["wd"])
    assert np.array_equal(UC.cosine_scheduler_step(0.99, 1, 39100, 0)[idx], G["ema"])
    assert UC.bool_flag("On") is True and UC.bool_flag("0") is False
    with pytest.raises(Exception):
        UC.bool_flag("maybe")


def test_view_grouping_and_padding():
    assert E.group_views([1001, 1001, 101, 101, 101, 101]) == [(0, 2), (2, 6)] == O.group_views([1001, 1001, 101, 101, 101, 101])
    assert E.group_views([1001]) == [(0, 1)] and E.group_views([5, 7, 7, 5]) == [(0, 1), (1, 3), (3, 4)]
    assert [E.pad_tokens(n) for n in (1, 26, 32, 33, 126, 250, 251, 256)] == [32, 32, 32, 64, 128, 256, 256, 256]
    with pytest.raises(Exception):
        E.pad_tokens(257)


def test_mask_samplers_match_oracle_restatement():
    a, b = np.random.RandomState(7), np.random.RandomState(7)
    for _ in range(20):
        assert np.array_equal(RM.block_mask(250, 0.65, 5, rng=a), O.block_mask(250, 0.65, 5, rng=b))
    torch.manual_seed(3)
    m = RM.get_mask_batch(4, 250, 0.65)
    assert m.shape == (4, 250) and m.dtype == torch.bool and int(m[0].sum()) == 163
    one = RM.get_mask_one(250, 200, 0.5)
    assert one.shape == (250,) and bool(one[200:].all()) and int(one[:200].sum()) == 100


def test_random_resize_crop_parameter_sampling():
    """The draws of BatchRandomResizeCrop follow get_params (transforms/byol_a.py:24-31): h, w from the scale ranges clipped
    to the canvas, (i, j) uniform over the positions that keep the crop inside the canvas."""
    import numpy as np
    from audiossl_amd.transforms import BatchRandomResizeCrop
    rrc = BatchRandomResizeCrop((1.0, 1.5), (0.6, 1.5), (0.6, 1.5), rng=np.random.RandomState(3))
    H, W = 64, 401
    CH, CW = rrc.canvas(H, W)
    assert (CH, CW) == (64, 601)
    P = rrc.sample_params(4000, H, W)
    i, j, h, w = P[:, 0], P[:, 1], P[:, 2], P[:, 3]
    assert h.min() >= int(0.6 * H) and h.max() == CH                       # freq scale up to 1.5 is clipped to the 64-row canvas
    assert w.min() >= int(0.6 * W) and w.max() <= CW and w.max() > W
    assert (i >= 0).all() and (i + h <= CH).all() and (j >= 0).all() and (j + w <= CW).all()
    assert (i[h == CH] == 0).all()
    assert abs(float((h == CH).mean()) - (1.5 - 1.0) / (1.5 - 0.6)) < 0.03      # P(U(0.6,1.5) * 64 >= 64)


# ---- ATST-Frame host surface (methods/atstframe/transform.py:14-101, data.py:20-111) ----------------------------------
def test_frame_transform_batch_contract():
    from audiossl_amd.methods.atstframe.transform import FrameATSTTrainTransform, get_num_patches
    from audiossl_amd.methods.atstframe.data import FrameATSTDataModule
    assert get_num_patches(64, 1001, 64, 4) == 250 and get_num_patches(64, 601, 64, 4) == 150
    t = FrameATSTTrainTransform(win_length=640, aug_tea=False, aug_stu=True, mask_ratio=0.65, mask_type="block", anchor_len=10, mask_len=5)
    wave = 0.1 * torch.randn(1, 170000)
    crops, lengths, masks = t(wave)
    assert len(crops) == 1 and crops[0].shape == (1, 160000)               # ONE crop feeds both views (transform.py:78-82)
    assert lengths == [1001, 1001]
    assert len(masks) == 2 and masks[0] is masks[1] and masks[0].shape == (250,) and masks[0].dtype == torch.bool
    assert 60 <= int(masks[0].sum()) <= 163                                # overlapping spans: realised ratio < 0.65
    short = t(0.1 * torch.randn(1, 100000))[0][0]                          # shorter than the crop: zero-padded (RandomCrop)
    assert short.shape == (1, 160000) and float(short[0, 100000:].abs().max()) == 0.0
    r = FrameATSTTrainTransform(mask_type="random", mask_ratio=0.75, anchor_len=6.)
    _, ln, mk = r(wave)
    assert ln == [601, 601] and mk[0].shape == (150,) and int(mk[0].sum()) == 113       # randperm < 150*0.75 -> 113 exactly
    u = FrameATSTTrainTransform(mask_type="uniform", mask_ratio=0.65, anchor_len=10, min_mask_len=2)
    assert u(wave)[2][0].shape == (250,)
    with pytest.raises(NotImplementedError):
        FrameATSTTrainTransform(n_mels=128)                                # patch_h must follow n_mels (one patch row: train.py:15 spec_h = n_mels)
    with pytest.raises(NotImplementedError):
        FrameATSTTrainTransform(n_mels=80, patch_h=80)
    # the reference's sr / n_mels / patch_h / patch_w parameters (transform.py:14-17, train.py:15,50-51) at BASELINE.json configs[4]'s geometry:
    # 32 kHz, 128 bands, 128 x 8 patches; anchor_len counts units of 16000 samples (transform.py:76), so 10 s at 32 kHz is anchor_len 20
    hi = FrameATSTTrainTransform(sr=32000, n_mels=128, patch_h=128, patch_w=8, win_length=1024, mask_type="block", mask_ratio=0.65, anchor_len=20, mask_len=5)
    crops, lengths, masks = hi(0.1 * torch.randn(1, 330000))
    assert crops[0].shape == (1, 320000) and lengths == [2001, 2001] and masks[0].shape == (250,) and get_num_patches(128, 2001, 128, 8) == 250
    # collate: what training_step receives -- ((crops, lengths, masks), label), ref: atstframe/model.py:120
    dm = FrameATSTDataModule(batch_size_per_gpu=4, num_workers=0, subset=8, win_length=640, aug_tea=False, mask_ratio=0.65,
                             anchor_len=10, mix_up=False)
    assert dm.transform.mix_up is True                                     # reference quirk: mix_up is not forwarded (data.py:46-57)
    (crops, lengths, masks), label = next(iter(dm.train_dataloader()))
    assert crops[0].shape == (4, 1, 160000) and [tuple(l.shape) for l in lengths] == [(4,), (4,)] and int(lengths[0][0]) == 1001
    assert masks[0].shape == (4, 250) and torch.equal(masks[0], masks[1]) and not masks[0].is_cuda
    assert label.shape == (4, 1, 527)
    bv = dm.transform.batch_views.__self__                                 # the GPU stage is configured from the transform
    assert bv.win_length == 640 and bv.aug_tea is False and bv.aug_stu is True


def test_block_mask_uniform_type_follows_fairseq_draw_order():
    """uniform: rand() (span count), randint(other, 2L+1, num) (lengths), choice(S - min_len, num) (starts)."""
    S, p, L, other = 250, 0.65, 5, 2
    rs = np.random.RandomState(9)
    got = RM.block_mask(S, p, L, rng=rs, mask_type="uniform", mask_other=other)
    rs = np.random.RandomState(9)
    num = max(2, int(p * S / float(L) + rs.rand()))
    lengths = rs.randint(other, 2 * L + 1, size=num)
    starts = rs.choice(S - int(lengths.min()), num, replace=False)
    want = np.zeros(S, bool)
    for s0, ln in zip(starts, lengths):
        want[s0:min(S, s0 + ln)] = True
    assert np.array_equal(got, want)


def test_lmdb_dataset_semantics_on_a_dict_store():
    """Subset shuffle / cycle window / item contract of audiossl/datasets/lmdb.py:12-98, key-value layer replaced."""
    import random
    from audiossl_amd.datasets import DictStore, LMDBDataset
    items = {f"clip{i:03d}".encode("ascii"): (np.full((1, 1, 1600), i, np.float32), np.eye(527, dtype=np.float32)[None, i % 527])
             for i in range(10)}
    random.seed(1234)
    ds = LMDBDataset("unused", "train", subset=4, store=DictStore(items), return_key=True)
    random.seed(1234)
    keys = list(items)
    random.shuffle(keys)
    assert ds.keys == keys[:4] and len(ds) == 4 and ds.num_classes == 527 and ds.start == 4
    wave, label, key = ds[0]
    assert wave.shape == (1, 1600) and label.shape == (527,) and key == keys[0] and float(wave[0, 0]) == int(keys[0][4:])
    ds.cycle()
    assert ds.keys == keys[4:8] and ds.start == 8
    ds.cycle()                                                             # wraps: tail + head, then reshuffles the pool
    assert ds.keys == keys[8:] + keys[:2] and ds.start == 0
    full = LMDBDataset("unused", "train", subset=None, store=DictStore(items), transform=lambda w: (w * 2, 7))
    (w2, seven), lab = full[3]
    assert len(full) == 10 and seven == 7 and float(w2[0, 0]) == 6.0
    with pytest.raises(RuntimeError):
        LMDBDataset("/nonexistent", "train")                               # no lmdb / legacy pyarrow in this image: loud failure


def test_bench_refuses_a_world_size_that_is_not_gpus():
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "4"], cwd=root, env=dict(os.environ, WORLD_SIZE="2", RANK="0"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and "WORLD_SIZE=2" in r.stderr


def test_cnn_patch_embed_layout_matches_reference_names():
    """patch_embed="CNN" (atstframe/audio_transformer.py:57-74,117-118): same flat layout, Conv2d names / shapes."""
    G = np.load(os.path.join(GOLD, "frame_cnn_patch_embed.npz"))
    shapes = E.encoder_param_shapes("small", None, True, patch_embed="CNN")
    assert [n for n, _ in shapes] == list(G["param_names"])
    assert [" ".join(str(d) for d in s) for _, s in shapes] == list(G["param_shapes"])
    L = E.FlatLayout("small", None, True, "CNN")
    Ll = E.FlatLayout("small", None, True, "Linear")
    assert L.entries["encoder.patch_embed.proj.weight"][0] == Ll.entries["encoder.patch_embed.patch_embed.weight"][0]
    assert L.n_student == Ll.n_student


def test_block_mask_against_the_hf_port_of_fairseq_compute_mask_indices():
    """fairseq is absent, so the block-mask sampler is a restatement of its published algorithm ("parity unpinned").  An independent
    cross-check: transformers' wav2vec2 `_compute_mask_indices` is a third-party port of the same fairseq function (static spans, overlap
    allowed).  Its start range is one longer than fairseq's (S - L + 1 against S - L), so the draws are not comparable bit for bit; what must
    agree: the span COUNT for the same numpy seed (both take it from the first uniform draw: int(p S / L + u), min_masks), that every
    maximal run is built from spans of length L, and the distribution of the masked fraction (mean over 2000 draws within 1 %)."""
    import numpy as np
    from transformers.models.wav2vec2.modeling_wav2vec2 import _compute_mask_indices
    from audiossl_amd.methods.atstframe.random_mask import block_mask
    S, p, L = 250, 0.65, 5
    frac_mine, frac_hf = [], []
    for seed in range(2000):
        np.random.seed(seed)
        u = np.random.rand()
        n_spans = max(2, int(p * S / L + u))
        np.random.seed(seed)
        mine = block_mask(S, p, L, min_masks=2)
        np.random.seed(seed)
        hf = _compute_mask_indices((1, S), p, L, min_masks=2)[0]
        assert mine.dtype == bool and mine.shape == (S,) and hf.shape == (S,)
        # both are unions of n_spans spans of length L: between L (all coincide) and n_spans * L masked frames, never a run shorter than L
        for m in (mine, hf):
            assert L <= int(m.sum()) <= n_spans * L
            runs = np.diff(np.flatnonzero(np.diff(np.concatenate([[0], m.astype(np.int8), [0]])) != 0))[::2]
            assert runs.min() >= L or (m[-1] and runs[-1] < L)              # only HF's clamp at the last frame can cut a run short
        assert not mine[S - 1] or mine[S - L:].all()                        # fairseq's start range ends at S - L - 1: frame S - 1 is only ever the tail of a full span
        frac_mine.append(mine.mean()); frac_hf.append(hf.mean())
    assert abs(np.mean(frac_mine) - np.mean(frac_hf)) < 0.01 * np.mean(frac_hf), (np.mean(frac_mine), np.mean(frac_hf))
    assert abs(np.std(frac_mine) - np.std(frac_hf)) < 0.15 * np.std(frac_hf)


def test_hf_adamw_restatement_against_torch_adamw():
    """transformers < 5's AdamW is gone from the image ("parity unpinned": oracle.hf_adamw_step follows its published algorithm).  Independent
    cross-check against torch.optim.AdamW, which differs from it in exactly two documented ways: (a) eps is added to sqrt(v) BEFORE the bias
    correction in HF (an effective eps / sqrt(1 - beta2^t)), (b) HF decays the weights after the Adam update, torch before.  With gradients
    bounded away from zero both effects are second order: over 6 steps the two trajectories must agree to 1e-6 while the parameters move by 6e-3,
    and with wd = 0 and eps -> 0 they must agree to fp32 rounding."""
    import torch
    from oracle import atst_oracle as O
    g = torch.Generator().manual_seed(0)
    p0 = torch.randn(4096, generator=g)
    grads = [torch.sign(torch.randn(4096, generator=g)) * (0.5 + torch.randn(4096, generator=g).abs()) for _ in range(6)]
    for lr, wd, eps, tol in ((1e-3, 0.04, 1e-6, 1e-6), (1e-3, 0.0, 1e-12, 2e-7)):
        p = p0.clone(); m = torch.zeros_like(p); v = torch.zeros_like(p)
        q = torch.nn.Parameter(p0.clone())
        opt = torch.optim.AdamW([q], lr=lr, betas=(0.9, 0.999), eps=eps, weight_decay=wd)
        for t, gr in enumerate(grads, 1):
            O.hf_adamw_step(p, gr, m, v, t, lr, wd, eps=eps)
            q.grad = gr.clone(); opt.step()
        moved = float((p - p0).abs().max())
        diff = float((p - q.detach()).abs().max())
        assert moved > 5e-3 and diff < tol, (lr, wd, eps, moved, diff)


def test_frame_row_bookkeeping_on_the_host():
    """ATST-Frame's per-step host bookkeeping (AtstEngine._valid / _frame_rows / _host_cat; numpy since round 6: CPU torch ops there stalled in the intra-op
    thread pool on busy hosts) against a plain-Python restatement of the reference's selection: the masked frames that are real frames, in (b, n) order, as flat
    row indices s * stride + n; rowflag = the mask image on the padded / packed row grid.  ref: methods/atstframe/audio_transformer.py:183-207 (mask_index
    applied to the valid frames), models/atst/audio_transformer.py:70-71 (patch_length)."""
    import types
    rng = np.random.default_rng(5)
    fake = types.SimpleNamespace(patch_w=4, device=torch.device("cpu"), upload=lambda t: t)
    for S, n_tok, stride in ((7, 250, 256), (64, 25, 25), (3, 100, 128)):
        width = n_tok * 4 + 1
        lengths = torch.from_numpy(rng.integers(5, width + 1, size=S)); lengths[0] = width
        valid = E.AtstEngine._valid(fake, lengths, 0, n_tok)
        want_valid = [min((int(l) - int(l) % 4) // 4, n_tok) for l in lengths]
        assert valid.dtype == torch.int32 and valid.tolist() == want_valid
        assert E.AtstEngine._valid(fake, lengths, 1, n_tok + 1).tolist() == [v + 1 for v in want_valid]          # clip encoders: + CLS
        mk = torch.from_numpy(rng.random((S, n_tok)) < 0.6)
        rows, rowflag = E.AtstEngine._frame_rows(fake, mk, valid, stride, True)
        want_rows = [s * stride + n for s in range(S) for n in range(n_tok) if bool(mk[s, n]) and n < want_valid[s]]
        assert rows.dtype == torch.int32 and rows.tolist() == want_rows
        rf = rowflag.reshape(S, stride)
        assert rf.dtype == torch.uint8 and torch.equal(rf[:, :n_tok].bool(), mk) and int(rf[:, n_tok:].sum()) == 0
        rows2, none = E.AtstEngine._frame_rows(fake, mk, valid, stride, False)
        assert none is None and torch.equal(rows2, rows)
    a, b = torch.arange(5), torch.arange(5, 9)
    assert torch.equal(E._host_cat([a, b]), torch.cat([a, b])) and E._host_cat([a]) is a
    m = [torch.from_numpy(rng.random((4, 10)) < 0.5) for _ in range(3)]
    assert torch.equal(E._host_cat(m), torch.cat(m)) and E._host_cat(m).dtype == torch.bool
