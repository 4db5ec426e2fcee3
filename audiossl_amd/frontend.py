"""Batched log-mel front end on the GPU (HIP kernel) with the numerics of the reference's ``mel_feature`` transform:
MelSpectrogram(sr, n_fft 1024, win W, hop 160, n_mels, 60-7800 Hz) -> AmplitudeToDB(power, top_db 80) -> MinMax.
ref: audiossl/methods/atst/transform.py:14-33 ; audiossl/methods/atstframe/transform.py:14-41 (win_length 640 recipe; `sr` and
`n_mels` are constructor parameters there: 16 kHz / 64 bands in the shipped recipes, 32 kHz / 128 bands = BASELINE.json configs[4])."""
from __future__ import annotations

import math

import torch

from . import hip

N_FFT, HOP, N_MELS, SR = 1024, 160, 64, 16000
F_MIN, F_MAX = 60.0, 7800.0


def _filterbank(sr: int = SR, n_mels: int = N_MELS) -> torch.Tensor:
    """HTK triangular filters [513, n_mels], evaluated in fp32 torch ops exactly like torchaudio.functional.melscale_fbanks."""
    all_freqs = torch.linspace(0, sr // 2, N_FFT // 2 + 1)
    m_min = 2595.0 * math.log10(1.0 + F_MIN / 700.0)
    m_max = 2595.0 * math.log10(1.0 + F_MAX / 700.0)
    m_pts = torch.linspace(m_min, m_max, n_mels + 2)
    f_pts = 700.0 * (10.0 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    down = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return torch.clamp(torch.min(down, up), min=0.0)


class LogMelFrontend:
    """``mel_feature`` for a batch of equal-length waveforms: wave [B, L] f32 (device) -> [B, 1, n_mels, 1 + L // 160].
    Rows may be slices of a longer buffer (row stride > L): the kernel takes the stride, nothing is copied.  ``out=`` lets
    consecutive views land in one [V * B, 1, n_mels, T] buffer, which the engine then uses without a torch.cat."""

    def __init__(self, win_length: int = 1024, device=None, sr: int = SR, n_mels: int = N_MELS):
        hip.load()
        if n_mels not in (64, 128):
            raise hip.HipError("the HIP mel front end is compiled for 64 or 128 bands")
        self.sr, self.n_mels = sr, n_mels
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.win_length = win_length
        w = torch.hann_window(win_length, periodic=True)
        if win_length < N_FFT:                                    # torch.stft pads the window centred to n_fft
            left = (N_FFT - win_length) // 2
            w = torch.nn.functional.pad(w, (left, N_FFT - win_length - left))
        self.window = w.to(self.device).contiguous()
        fb = _filterbank(sr, n_mels)                              # compact: per band first bin, length, weights
        nz = fb > 0
        start = torch.tensor([int(nz[:, m].nonzero()[0]) if nz[:, m].any() else 0 for m in range(n_mels)])
        end = torch.tensor([int(nz[:, m].nonzero()[-1]) + 1 if nz[:, m].any() else 0 for m in range(n_mels)])
        self.maxlen = max(int((end - start).max()), 1)
        wts = torch.zeros(n_mels, self.maxlen)
        for m in range(n_mels):
            wts[m, : end[m] - start[m]] = fb[start[m]:end[m], m]
        self.fb_w = wts.t().contiguous().to(self.device)          # [tap, band]: the kernel reads one coalesced row of all bands per tap
        self.fb_start = start.to(torch.int32).to(self.device)
        self.fb_len = (end - start).to(torch.int32).to(self.device)

    def __call__(self, wave: torch.Tensor, out: torch.Tensor = None) -> torch.Tensor:
        if wave.dim() == 3:                                       # [B,1,L] as produced by the reference's datasets
            wave = wave[:, 0]
        wave = wave.to(self.device, torch.float32)
        if wave.stride(1) != 1 or (wave.shape[0] > 1 and wave.stride(0) < wave.shape[1]):
            wave = wave.contiguous()
        B, L = wave.shape
        ld = wave.stride(0) if B > 1 else L
        T = 1 + L // HOP
        if out is None:
            out = torch.empty(B, 1, self.n_mels, T, device=self.device)
        assert out.shape == (B, 1, self.n_mels, T) and out.is_contiguous() and out.dtype == torch.float32
        clipmax = torch.empty(B, dtype=torch.int32, device=self.device)
        hip.call("atst_mel_frontend_f32", wave.data_ptr(), B, L, ld, self.n_mels, self.win_length, hip.ptr(self.window), hip.ptr(self.fb_w),
                 hip.ptr(self.fb_start), hip.ptr(self.fb_len), self.maxlen, hip.ptr(out), hip.ptr(clipmax), hip.stream())
        return out
