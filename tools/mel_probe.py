#!/usr/bin/env python3
"""Mel front end stand-alone: 512 clip-views of 10 s @ 16 kHz (the global-view group of one step) and 1024 local views of 1 s."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audiossl_amd.frontend import LogMelFrontend
def t_us(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
for sr, nm, label in ((16000, 64, "16 kHz / 64 mel"), (32000, 128, "32 kHz / 128 mel")):
    fe = LogMelFrontend(win_length=1024, sr=sr, n_mels=nm)
    w = torch.randn(512, 10 * sr, device="cuda") * 0.1
    print(f"{label}: 512 x 10 s  {t_us(lambda: fe(w)):8.1f} us", flush=True)
    w = torch.randn(1024, sr, device="cuda") * 0.1
    print(f"{label}: 1024 x 1 s  {t_us(lambda: fe(w)):8.1f} us", flush=True)
