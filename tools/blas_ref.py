#!/usr/bin/env python3
"""Reference point only: library GEMM (torch.matmul -> hipBLASLt/rocBLAS) on the encoder's GEMM shapes, plain bf16 output."""
import torch
def t_us(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
M = 131072
for N, K, name in [(1152, 384, "qkv fwd"), (384, 384, "proj"), (1536, 384, "fc1"), (384, 1536, "fc2 / fc1 dgrad"), (384, 1152, "qkv dgrad")]:
    A = torch.randn(M, K, device="cuda").bfloat16(); B = torch.randn(N, K, device="cuda").bfloat16()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    t = t_us(lambda: torch.matmul(A, B.t(), out=out))
    print(f"  blas nt {name:18s} N={N:5d} K={K:5d} {t:8.1f} us  {2.0 * M * N * K / t / 1e6:7.1f} TF/s")
for N, K, name in [(384, 1536, "fc2 wgrad"), (1536, 384, "fc1 wgrad"), (384, 384, "proj wgrad"), (1152, 384, "qkv wgrad")]:
    dY = torch.randn(M, N, device="cuda").bfloat16(); X = torch.randn(M, K, device="cuda").bfloat16()
    out = torch.empty(N, K, device="cuda", dtype=torch.bfloat16)
    t = t_us(lambda: torch.matmul(dY.t(), X, out=out))
    print(f"  blas tn {name:18s} N={N:5d} K={K:5d} {t:8.1f} us  {2.0 * M * N * K / t / 1e6:7.1f} TF/s")
