"""Size-independent properties at the benchmark's full size (BASELINE.json configs[1]: 256 clips per GPU, 2 global +
4 local views), where the oracle is too slow to serve as the checker:
  * permutation equivariance: permuting the clips permutes the per-clip encoder outputs (DropPath off);
  * padding: the tile-padding token rows (251..255) enter the encoder as exact zeros (they are never keys, so whatever
    the blocks then write into them cannot reach a real token);
  * linearity of the backward pass in the upstream gradient scale;
  * EMA end points: m = 1 leaves the teacher untouched, m = 0 copies the student;
  * the loss stays inside the cosine-loss range and is finite for every view count."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from audiossl_amd.engine import AtstEngine  # noqa: E402
from oracle import atst_oracle as O  # noqa: E402

B = 256


def _views(seed, ncrops):
    g = torch.Generator().manual_seed(seed)
    mels = [torch.randn(B, 1, 64, 1001, generator=g).clamp_(-1, 1) for _ in range(2)]
    lens = [torch.full((B,), 1001), torch.randint(400, 1002, (B,), generator=g)]
    if ncrops == 6:
        mels += [torch.randn(B, 1, 64, 101, generator=g).clamp_(-1, 1) for _ in range(4)]
        lens += [torch.full((B,), 101)] * 4
    return mels, lens


def test_permutation_equivariance_and_padding_full_size():
    eng = AtstEngine("small", drop_path_rate=0.0)
    eng.init_weights(seed=3)
    mels, lens = _views(11, 2)
    S = 2 * B
    mel = torch.cat(mels).cuda(); length = torch.cat(lens)
    ep = eng._pass("student", S, 1001, True, 0)
    out = ep.forward(mel, eng._valid(length, 1), None, None).float().reshape(S, 256, 384)
    cls = out[:, 0].clone()
    tok = ep.tokens().reshape(S, 256, 384)
    # rows 251..255 are the tile padding (10 s -> 250 patches + CLS); rows between a clip's valid length and 251 are real
    # tokens of the zero-padded batch tensor: never keys, but computed as queries exactly as in the reference
    pad = (torch.arange(256)[None, :] >= 251).expand(S, 256)
    assert float(tok[pad.cuda()].abs().max()) == 0.0                              # pad rows: exact zeros at the input
    perm = torch.randperm(S, generator=torch.Generator().manual_seed(5))
    out2 = ep.forward(mel[perm.cuda()].contiguous(), eng._valid(length[perm], 1), None, None).float().reshape(S, 256, 384)
    d = (out2[:, 0] - cls[perm.cuda()]).abs().max() / cls.abs().max()
    assert float(d) < 1e-6, float(d)                                               # same kernels, same per-sequence arithmetic


@pytest.mark.parametrize("ncrops", [2, 6])
def test_backward_linearity_loss_range_and_ema_endpoints(ncrops):
    eng = AtstEngine("small", ncrops=ncrops, drop_path_rate=0.0)
    eng.init_weights(seed=4)
    mels, lens = _views(13, ncrops)
    loss, std_s, std_t = eng.forward(mels, lens)
    assert torch.isfinite(loss) and 0.0 <= float(loss) <= 4.0                     # 2 - 2 cos in [0, 4]
    assert 0.0 < float(std_s) < 0.2 and 0.0 < float(std_t) < 0.2                   # std of unit-norm 256-d rows <= 1/16
    eng.backward(grad_scale=1.0)
    g1 = eng.g32.clone()
    assert torch.isfinite(g1).all()
    eng.forward(mels, lens)
    eng.backward(grad_scale=2.0)
    rel = float((eng.g32 - 2.0 * g1).norm() / (2.0 * g1.norm()))
    assert rel < 2e-3, rel                                                         # fp32 atomics reorder sums; bf16 cast of the scaled upstream
    t_before = eng.t32.clone()
    eng.ema_update(1.0)
    assert torch.equal(eng.t32, t_before)
    eng.ema_update(0.0)
    for name in eng.layout.entries:
        if not name.startswith("predictor."):
            assert torch.equal(eng.param_view("teacher", name), eng.param_view("student", name)), name


def test_frame_full_size_properties():
    """ATST-Frame at the benchmark size (256 clips x 2 views, 250 frames, block masks 0.65 / 5): the head sees exactly the
    masked valid frames; permuting the clips (with their masks) leaves the loss unchanged; masks on the host and masks
    on the device give the same rows; backward is linear in the upstream scale; gradients reach mask_embed."""
    from audiossl_amd.methods.atstframe.random_mask import block_mask
    eng = AtstEngine("small", frame=True, drop_path_rate=0.0)
    eng.init_weights(seed=6)
    g = torch.Generator().manual_seed(17)
    mels = [torch.randn(B, 1, 64, 1001, generator=g).clamp_(-1, 1) for _ in range(2)]
    lens = [torch.randint(400, 1002, (B,), generator=g)] * 2
    rs = np.random.RandomState(3)
    m = torch.from_numpy(np.stack([block_mask(250, 0.65, 5, rng=rs) for _ in range(B)]))
    loss, std_s, std_t = eng.forward(mels, lens, [m, m])
    plen = (lens[0] - lens[0] % 4) // 4
    want_rows = int((m & (torch.arange(250)[None, :] < plen[:, None])).sum()) * 2
    assert sum(r.numel() for _, r in eng._student_groups) == want_rows            # bit-exact ragged row count
    rows_host = eng._student_groups[0][1].clone()
    assert torch.isfinite(loss) and 0.0 <= float(loss) <= 4.0
    eng.backward()
    g1 = eng.g32.clone()
    assert torch.isfinite(g1).all() and float(eng.param_view("student", "encoder.mask_embed", grad=True).abs().max()) > 0
    loss_dev, _, _ = eng.forward(mels, lens, [m.cuda(), m.cuda()])                 # device-side masks: same rows, same loss
    assert torch.equal(eng._student_groups[0][1], rows_host) and abs(float(loss_dev) - float(loss)) < 1e-6
    eng.backward(grad_scale=2.0)
    rel = float((eng.g32 - 2.0 * g1).norm() / (2.0 * g1.norm()))
    assert rel < 2e-3, rel
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(9))
    loss_p, _, _ = eng.forward([x[perm] for x in mels], [l[perm] for l in lens], [m[perm], m[perm]])
    assert abs(float(loss_p) - float(loss)) < 2e-4, (float(loss_p), float(loss))   # BatchNorm sums reorder: fp32 noise only


def test_training_reduces_the_loss_end_to_end():
    """30 optimizer steps on one fixed batch (12 layers, 16 clips x 2 views, DropPath on, HF-AdamW + EMA teacher through the
    fused kernels): the BYOL loss must fall -- forward, backward, optimizer, bf16 shadow refresh and EMA are consistent."""
    eng = AtstEngine("small", drop_path_rate=0.1)
    eng.init_weights(seed=8)
    g = torch.Generator().manual_seed(21)
    base = torch.randn(16, 1, 64, 1001, generator=g).clamp_(-1, 1)
    mels = [(base + 0.05 * torch.randn(16, 1, 64, 1001, generator=g)).clamp_(-1, 1).cuda() for _ in range(2)]
    lens = [torch.full((16,), 1001)] * 2
    torch.manual_seed(0)
    losses = []
    for k in range(30):
        loss, _, _ = eng.forward(mels, lens)
        eng.backward()
        eng.optimizer_step(5e-4, 0.04, 0.99)
        losses.append(loss)
    losses = [float(l) for l in losses]
    assert all(np.isfinite(losses)) and np.mean(losses[-5:]) < np.mean(losses[:5]) - 0.05, losses
    assert torch.isfinite(eng.p32).all() and torch.isfinite(eng.t32).all()
