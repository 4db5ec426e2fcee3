#!/usr/bin/env python3
"""Streaming bandwidth reference points on the box (torch fill / copy / add), to price epilogue stores against."""
import torch
def t_us(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for mb in (100, 300, 800):
    n = mb * 1024 * 1024 // 4
    x = torch.empty(n, device="cuda"); y = torch.empty(n, device="cuda"); z = torch.empty(n, device="cuda")
    t = t_us(lambda: x.zero_()); print(f"{mb:4d} MB fill   {t:8.1f} us  {mb * 1.048576 / t:6.2f} TB/s written")
    t = t_us(lambda: y.copy_(x)); print(f"{mb:4d} MB copy   {t:8.1f} us  {2 * mb * 1.048576 / t:6.2f} TB/s r+w")
    t = t_us(lambda: torch.add(x, y, out=z)); print(f"{mb:4d} MB add    {t:8.1f} us  {3 * mb * 1.048576 / t:6.2f} TB/s 2r+w")
    t = t_us(lambda: x.sum()); print(f"{mb:4d} MB sum    {t:8.1f} us  {mb * 1.048576 / t:6.2f} TB/s read")
