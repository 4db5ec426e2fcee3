#!/usr/bin/env python3
"""Head-Linear shapes (fp32 output, few tiles, long K): one block per tile (hook 380) against split-K (381).  GPU box."""
import os, sys, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audiossl_amd import hip
lib = hip.load()
for M, N, K in ((1536, 256, 12288), (512, 256, 12288), (1536, 384, 4096), (1536, 256, 4096), (512, 384, 4096)):
    A = torch.randn(M, K, device="cuda").bfloat16(); B = (torch.randn(N, K, device="cuda") * 0.05).bfloat16(); out = torch.empty(M, N, device="cuda")
    f = lambda: hip.call("atst_gemm_nt_bf16", hip.ptr(A), hip.ptr(B), M, N, K, K, K, hip.EPI_F32, hip.ptr(out), N, None, None, None, None, 1, None, None, None, None, None, hip.stream())
    t = {}
    for v in (0, 1, 0, 1):
        lib.atst_tune_gemm_variant(380 + v)
        f(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize()
        t.setdefault(v, []).append(e0.elapsed_time(e1) / 20 * 1e3)
    print(f"M={M:5d} N={N:4d} K={K:6d}  one block per tile {min(t[0]):7.1f} us   split-K {min(t[1]):7.1f} us")
lib.atst_tune_gemm_variant(381)
