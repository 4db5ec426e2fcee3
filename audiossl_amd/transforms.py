"""Batched (device-side) versions of the reference's host augmentations, applied after collate on [B,1,64,T] log-mel
batches: BYOL-A ``Mixup`` (log-mix-exp against a FIFO memory bank) and ``RandomResizeCrop`` (virtual canvas + random
crop + bicubic resize), plus ``MinMax`` / ``RandomCrop`` restated from audiossl/transforms/common.py:63-74,97-110.
ref: audiossl/transforms/byol_a.py:7-49 (RandomResizeCrop), :61-115 (log_mixup_exp, Mixup).

These are stochastic augmentations: the distributions match the reference (same parameter sampling per sample); the
random streams do not (numpy per-worker RNG there, one torch CPU generator here).  One deliberate difference: the resize
samples the canvas with ``grid_sample(bicubic, align_corners=True)``, which reads real canvas pixels just outside the
crop where ``F.interpolate`` on the cropped tensor replicates the crop's border (<= 2 pixels at the crop edge)."""
from __future__ import annotations

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


class BatchMixup:
    """Mixup(ratio=0.4, n_memory=2000, log_mixup_exp=True) over a batch; the bank holds past *un-mixed* inputs."""

    def __init__(self, ratio=0.4, n_memory=2000, generator: torch.Generator = None):
        self.ratio, self.n, self.gen = ratio, n_memory, generator
        self.bank = None          # [n_filled, 1, H, T] on the batch's device
        self.filled = 0
        self.head = 0

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        B = x.shape[0]
        out = x
        if self.filled > 0 and self.bank.shape[-1] >= x.shape[-1]:
            alpha = (self.ratio * torch.rand(B, generator=self.gen)).to(x.device).view(B, 1, 1, 1)
            idx = torch.randint(0, self.filled, (B,), generator=self.gen).to(x.device)
            z = self.bank[idx]
            if z.shape[-1] > x.shape[-1]:                     # shorter input: random window of the bank entry
                s = int(torch.randint(0, z.shape[-1] - x.shape[-1], (1,), generator=self.gen))
                z = z[..., s:s + x.shape[-1]]
            mixed = (1.0 - alpha) * x.exp() + alpha * z.exp()
            out = torch.log(mixed + torch.finfo(x.dtype).eps)
        self._push(x)
        return out.float()

    def _push(self, x):
        if self.bank is None:
            self.bank = torch.zeros((self.n,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        if x.shape[1:] != self.bank.shape[1:]:
            return                                            # bank keeps one geometry (the first view length seen)
        for i in range(0, x.shape[0], self.n):
            chunk = x[i:i + self.n]
            k = chunk.shape[0]
            pos = (self.head + torch.arange(k, device=x.device)) % self.n
            self.bank[pos] = chunk
            self.head = (self.head + k) % self.n
            self.filled = min(self.n, self.filled + k)


class BatchRandomResizeCrop:
    """RandomResizeCrop(virtual_crop_scale, freq_scale, time_scale) with one parameter draw per sample."""

    def __init__(self, virtual_crop_scale=(1.0, 1.5), freq_scale=(0.6, 1.5), time_scale=(0.6, 1.5), generator=None):
        assert time_scale[1] >= 1.0 and freq_scale[1] >= 1.0
        self.vcs, self.fs, self.ts, self.gen = virtual_crop_scale, freq_scale, time_scale, generator

    def sample_params(self, B, H, W):
        CH, CW = int(H * self.vcs[0]), int(W * self.vcs[1])
        u = torch.rand(B, 4, generator=self.gen)
        h = (((self.fs[0] + (self.fs[1] - self.fs[0]) * u[:, 0]) * H).long()).clamp(1, CH)
        w = (((self.ts[0] + (self.ts[1] - self.ts[0]) * u[:, 1]) * W).long()).clamp(1, CW)
        i = (u[:, 2] * (CH - h + 1).float()).long().clamp(max=CH - 1)      # randint(0, CH-h) inclusive
        j = (u[:, 3] * (CW - w + 1).float()).long().clamp(max=CW - 1)
        return CH, CW, i, j, h, w

    def __call__(self, lms: torch.Tensor) -> torch.Tensor:
        B, C, H, W = lms.shape
        CH, CW, i, j, h, w = self.sample_params(B, H, W)
        canvas = torch.zeros(B, C, CH, CW, dtype=lms.dtype, device=lms.device)
        y0, x0 = (CH - H) // 2, (CW - W) // 2
        canvas[:, :, y0:y0 + H, x0:x0 + W] = lms
        dev = lms.device
        i, j, h, w = (t.to(dev).float().view(B, 1) for t in (i, j, h, w))
        ys = torch.linspace(0, 1, H, device=dev).view(1, H)
        xs = torch.linspace(0, 1, W, device=dev).view(1, W)
        sy = i + ys * (h - 1)                                  # align_corners=True mapping of the crop onto H x W
        sx = j + xs * (w - 1)
        gy = 2.0 * sy / max(CH - 1, 1) - 1.0
        gx = 2.0 * sx / max(CW - 1, 1) - 1.0
        grid = torch.stack((gx.view(B, 1, W).expand(B, H, W), gy.view(B, H, 1).expand(B, H, W)), dim=-1)
        return F.grid_sample(canvas, grid, mode="bicubic", padding_mode="border", align_corners=True).float()
