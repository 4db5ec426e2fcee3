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
