"""Batched log-mel front end on the GPU (HIP kernel) with the numerics of the reference's ``mel_feature`` transform:
MelSpectrogram(16 kHz, n_fft 1024, win W, hop 160, 64 mel, 60-7800 Hz) -> AmplitudeToDB(power, top_db 80) -> MinMax.
ref: audiossl/methods/atst/transform.py:14-33 ; audiossl/methods/atstframe/transform.py:16-41 (win_length 640 recipe)."""
from __future__ import annotations

import math

import torch

from . import hip

N_FFT, HOP, N_MELS, SR = 1024, 160, 64, 16000
F_MIN, F_MAX = 60.0, 7800.0


def _filterbank() -> torch.Tensor:
    """HTK triangular filters [513, 64], evaluated in fp32 torch ops exactly like torchaudio.functional.melscale_fbanks."""
    all_freqs = torch.linspace(0, SR // 2, N_FFT // 2 + 1)
    m_min = 2595.0 * math.log10(1.0 + F_MIN / 700.0)
    m_max = 2595.0 * math.log10(1.0 + F_MAX / 700.0)
    m_pts = torch.linspace(m_min, m_max, N_MELS + 2)
    f_pts = 700.0 * (10.0 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    down = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return torch.clamp(torch.min(down, up), min=0.0)


class LogMelFrontend:
    """``mel_feature`` for a batch of equal-length waveforms: wave [B, L] f32 (device) -> [B, 1, 64, 1 + L // 160]."""

    def __init__(self, win_length: int = 1024, device=None):
        hip.load()
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.win_length = win_length
        w = torch.hann_window(win_length, periodic=True)
        if win_length < N_FFT:                                    # torch.stft pads the window centred to n_fft
            left = (N_FFT - win_length) // 2
            w = torch.nn.functional.pad(w, (left, N_FFT - win_length - left))
        self.window = w.to(self.device).contiguous()
        fb = _filterbank()                                        # compact: per band first bin, length, weights
        nz = fb > 0
        start = torch.tensor([int(nz[:, m].nonzero()[0]) if nz[:, m].any() else 0 for m in range(N_MELS)])
        end = torch.tensor([int(nz[:, m].nonzero()[-1]) + 1 if nz[:, m].any() else 0 for m in range(N_MELS)])
        self.maxlen = int((end - start).max())
        wts = torch.zeros(N_MELS, self.maxlen)
        for m in range(N_MELS):
            wts[m, : end[m] - start[m]] = fb[start[m]:end[m], m]
        self.fb_w = wts.to(self.device).contiguous()
        self.fb_start = start.to(torch.int32).to(self.device)
        self.fb_len = (end - start).to(torch.int32).to(self.device)

    def __call__(self, wave: torch.Tensor) -> torch.Tensor:
        if wave.dim() == 3:                                       # [B,1,L] as produced by the reference's datasets
            wave = wave[:, 0]
        wave = wave.to(self.device, torch.float32).contiguous()
        B, L = wave.shape
        T = 1 + L // HOP
        out = torch.empty(B, 1, N_MELS, T, device=self.device)
        clipmax = torch.empty(B, dtype=torch.int32, device=self.device)
        hip.call("atst_mel_frontend_f32", hip.ptr(wave), B, L, self.win_length, hip.ptr(self.window), hip.ptr(self.fb_w),
                 hip.ptr(self.fb_start), hip.ptr(self.fb_len), self.maxlen, hip.ptr(out), hip.ptr(clipmax), hip.stream())
        return out
