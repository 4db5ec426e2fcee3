#!/usr/bin/env python3
"""e4m3 weight gradient (atst_gemm_tn_fp8) next to the bf16 one (atst_gemm_tn_bf16) on the ATST-base shapes at M = 131072 (run on the GPU box)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audiossl_amd import hip
hip.load(); dev = "cuda"; M = int(os.environ.get("M", 131072))
def t_us(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for name, N, K in (("fc1 wgrad", 3072, 768), ("fc2 wgrad", 768, 3072), ("proj wgrad", 768, 768), ("qkv wgrad", 2304, 768)):
    if N % 256 or K % 256: continue
    dY = (torch.randn(M, N, device=dev) * 0.02).bfloat16(); X = torch.randn(M, K, device=dev).bfloat16()
    dY8 = torch.empty(M, N, dtype=torch.uint8, device=dev); X8 = torch.empty(M, K, dtype=torch.uint8, device=dev)
    hip.call("atst_quant_fp8_bf16", hip.ptr(dY), M * N, 2048.0, hip.ptr(dY8), hip.stream()); hip.call("atst_quant_fp8_bf16", hip.ptr(X), M * K, 8.0, hip.ptr(X8), hip.stream())
    sy, sx = torch.tensor([2048.0], device=dev), torch.tensor([8.0], device=dev)
    dW = torch.zeros(N, K, device=dev)
    t16 = t_us(lambda: hip.call("atst_gemm_tn_bf16", hip.ptr(dY), hip.ptr(X), M, N, K, N, K, hip.ptr(dW), K, 0, hip.stream()))
    t8 = t_us(lambda: hip.call("atst_gemm_tn_fp8", hip.ptr(dY8), hip.ptr(X8), M, N, K, N, K, hip.ptr(dW), K, hip.ptr(sy), hip.ptr(sx), hip.stream()))
    fl = 2.0 * M * N * K
    print(f"  {name:12s} N={N:5d} K={K:5d}   bf16 {t16:8.1f} us {fl/t16/1e6:6.0f} TF   e4m3 {t8:8.1f} us {fl/t8/1e6:6.0f} TF   x{t16/t8:.2f}", flush=True)
