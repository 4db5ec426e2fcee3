#!/usr/bin/env python3
"""Does a tensor that was just written get read faster than a cold one (256 MB Infinity Cache)?  sizes in MB."""
import torch
def ev(): return torch.cuda.Event(enable_timing=True)
big = torch.empty(1536 * 1024 * 1024 // 4, device="cuda")
for mb in (32, 64, 100, 200, 400):
    n = mb * 1024 * 1024 // 4
    src = torch.empty(n, device="cuda"); dst = torch.empty(n, device="cuda")
    res = {}
    for mode in ("warm", "cold"):
        ts = []
        for _ in range(5):
            src.fill_(1.0)                      # producer
            if mode == "cold":
                big.fill_(0.0)                  # evict
            e0, e1 = ev(), ev()
            e0.record(); s = src.sum(); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        res[mode] = sorted(ts)[len(ts) // 2]
    print(f"{mb:4d} MB  read after write: warm {res['warm']:7.1f} us ({mb * 1.048576 / res['warm']:.2f} TB/s)   cold {res['cold']:7.1f} us ({mb * 1.048576 / res['cold']:.2f} TB/s)")
