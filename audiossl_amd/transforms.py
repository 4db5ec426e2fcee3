"""Batched (device-side) versions of the reference's host augmentations, applied after collate on [B,1,64,T] log-mel
batches: BYOL-A ``Mixup`` (log-mix-exp against a FIFO memory bank) and ``RandomResizeCrop`` (virtual canvas + random
crop + bicubic resize), plus ``MinMax`` / ``RandomCrop`` restated from audiossl/transforms/common.py:63-74,97-110.
ref: audiossl/transforms/byol_a.py:7-49 (RandomResizeCrop), :61-115 (log_mixup_exp, Mixup).

The arithmetic runs in HIP kernels (csrc/augment.hip) that take the random draws as inputs; given the same draws the
outputs equal the reference's (goldens: tests/golden/aug_byol_a.npz).  The distributions of the draws match the
reference (same per-sample sampling order); the random streams do not (numpy / python global RNGs per dataloader worker
there, one RandomState per transform object here)."""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F


class MinMax:
    """ref: transforms/common.py:97-110."""

    def __init__(self, min, max):
        self.min, self.max = min, max

    def __call__(self, x):
        lo, hi = (torch.min(x), torch.max(x)) if self.min is None else (self.min, self.max)
        return (x - lo) / (hi - lo) * 2.0 - 1.0


class RandomCrop:
    """ref: transforms/common.py:63-74 (waveform [1, N] -> [1, size], zero-padded when shorter)."""

    def __init__(self, size: int, pad: bool = True, generator: torch.Generator = None):
        self.size, self.pad, self.gen = size, pad, generator

    def __call__(self, signal):
        n = signal.shape[-1]
        if n < self.size:
            return F.pad(signal, (0, self.size - n)) if self.pad else signal
        start = int(torch.randint(0, n - self.size + 1, (1,), generator=self.gen))
        return signal[..., start:start + self.size]


def _i32(a, device):
    return torch.as_tensor(np.asarray(a, dtype=np.int32), device=device).contiguous()


class BatchMixup:
    """Mixup(ratio=0.4, n_memory=2000, log_mixup_exp=True) over a collated batch (ref: byol_a.py:86-115), executed by
    ``atst_log_mixup_exp_f32``.  Per sample, as in the reference: a = ratio * U(0,1); z = a uniformly drawn bank entry;
    out = log((1-a) e^x + a e^z + eps); when x and z differ in length the shorter one is mixed into a random window of
    the other (byol_a.py:66-77).  The bank holds past *un-mixed* inputs (FIFO).
    Differences from the per-item reference, both deliberate: the batch is appended to the bank after it has been
    mixed (there, item k already sees items < k of its own batch), and the bank keeps one geometry (the first width it
    sees; the reference keeps a heterogeneous python list).  The random stream is this object's RandomState, not numpy's
    global one."""

    def __init__(self, ratio=0.4, n_memory=2000, rng: "np.random.RandomState" = None):
        self.ratio, self.n = ratio, n_memory
        self.rng = rng if rng is not None else np.random.RandomState()
        self.bank = None          # [n, H, Wz] on the batch's device
        self.filled = 0
        self.head = 0

    def sample_params(self, B, W):
        """-> (alpha [B] f32, zidx, zstart, xstart [B] i32), drawn per sample in the reference's order."""
        Wz = self.bank.shape[-1]
        alpha = np.empty(B, np.float32); zidx = np.empty(B, np.int32)
        zstart = np.zeros(B, np.int32); xstart = np.zeros(B, np.int32)
        for b in range(B):
            alpha[b] = self.ratio * self.rng.random_sample()
            zidx[b] = self.rng.randint(self.filled)
            if W < Wz:
                zstart[b] = self.rng.randint(0, Wz - W)
            elif W > Wz:
                xstart[b] = self.rng.randint(0, W - Wz)
        return alpha, zidx, zstart, xstart

    def apply(self, x, alpha, zidx, zstart, xstart):
        from . import hip
        shape = x.shape
        x3 = x.reshape(-1, shape[-2], shape[-1]).float().contiguous()
        B, H, W = x3.shape
        out = torch.empty_like(x3)
        # bank slots are addressed through the ring: logical entry k (oldest first) lives at (head - filled + k) mod n
        phys = (self.head - self.filled + np.asarray(zidx, np.int64)) % self.n
        # keep the argument tensors alive until the launch has been enqueued (a temporary freed inside the call expression
        # hands its memory to the next temporary)
        d_idx, d_zs, d_xs = _i32(phys, x.device), _i32(zstart, x.device), _i32(xstart, x.device)
        d_alpha = torch.as_tensor(np.asarray(alpha, np.float32), device=x.device).contiguous()
        hip.call("atst_log_mixup_exp_f32", hip.ptr(x3), hip.ptr(self.bank), hip.ptr(d_idx), hip.ptr(d_zs), hip.ptr(d_xs),
                 hip.ptr(d_alpha), hip.ptr(out), B, H, W, self.bank.shape[-1], hip.stream())
        return out.reshape(shape)

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        out = x
        if self.filled > 0:
            B = x.reshape(-1, x.shape[-2], x.shape[-1]).shape[0]
            out = self.apply(x, *self.sample_params(B, x.shape[-1]))
        self._push(x)
        return out.float()

    def _push(self, x):
        x3 = x.reshape(-1, x.shape[-2], x.shape[-1]).float()
        if self.bank is None:
            self.bank = torch.zeros((self.n,) + tuple(x3.shape[1:]), dtype=torch.float32, device=x.device)
        if x3.shape[1:] != self.bank.shape[1:]:
            return                                            # bank keeps one geometry (the first view length seen)
        for i in range(0, x3.shape[0], self.n):
            chunk = x3[i:i + self.n]
            k = chunk.shape[0]
            pos = (self.head + torch.arange(k, device=x.device)) % self.n
            self.bank[pos] = chunk
            self.head = (self.head + k) % self.n
            self.filled = min(self.n, self.filled + k)


class BatchRandomResizeCrop:
    """RandomResizeCrop(virtual_crop_scale, freq_scale, time_scale) with one parameter draw per sample
    (ref: byol_a.py:7-49), executed by ``atst_rrc_bicubic_f32``: the zero canvas is never materialised, taps are clamped
    to the crop exactly as ``F.interpolate(crop, mode='bicubic', align_corners=True)`` does.  Given the same
    (i, j, h, w) the output equals the reference's (tests/golden/aug_byol_a.npz); the draws come from this object's
    RandomState instead of numpy's / python's global generators."""

    def __init__(self, virtual_crop_scale=(1.0, 1.5), freq_scale=(0.6, 1.5), time_scale=(0.6, 1.5), rng: "np.random.RandomState" = None):
        assert time_scale[1] >= 1.0 and freq_scale[1] >= 1.0
        self.vcs, self.fs, self.ts = virtual_crop_scale, freq_scale, time_scale
        self.rng = rng if rng is not None else np.random.RandomState()

    def canvas(self, H, W):
        return int(H * self.vcs[0]), int(W * self.vcs[1])

    def sample_params(self, B, H, W):
        """-> int32 [B,4] rows (i, j, h, w), drawn as get_params does (byol_a.py:24-31)."""
        CH, CW = self.canvas(H, W)
        out = np.empty((B, 4), np.int32)
        for b in range(B):
            h = int(np.clip(int(self.rng.uniform(*self.fs) * H), 1, CH))
            w = int(np.clip(int(self.rng.uniform(*self.ts) * W), 1, CW))
            i = self.rng.randint(0, CH - h + 1) if CH > h else 0
            j = self.rng.randint(0, CW - w + 1) if CW > w else 0
            out[b] = (i, j, h, w)
        return out

    def apply(self, lms: torch.Tensor, params) -> torch.Tensor:
        from . import hip
        shape = lms.shape
        x3 = lms.reshape(-1, shape[-2], shape[-1]).float().contiguous()
        B, H, W = x3.shape
        CH, CW = self.canvas(H, W)
        out = torch.empty_like(x3)
        d_params = _i32(params, lms.device)
        hip.call("atst_rrc_bicubic_f32", hip.ptr(x3), hip.ptr(out), hip.ptr(d_params), B, H, W, CH, CW, hip.stream())
        return out.reshape(shape)

    def __call__(self, lms: torch.Tensor) -> torch.Tensor:
        B = lms.reshape(-1, lms.shape[-2], lms.shape[-1]).shape[0]
        return self.apply(lms, self.sample_params(B, lms.shape[-2], lms.shape[-1]))
