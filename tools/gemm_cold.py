#!/usr/bin/env python3
"""fc1(+gelu) / qkv GEMM timing with hot (same buffers) vs cold (rotating buffer sets) operands."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audiossl_amd import hip
hip.load(); dev = "cuda"; M = 131072
def run(N, K, epi, nsets, label):
    sets = []
    for _ in range(nsets):
        A = torch.randn(M, K, device=dev).bfloat16(); B = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16); C2 = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        sets.append((A, B, out, C2))
    bias = torch.randn(N, device=dev)
    def call(i):
        A, B, out, C2 = sets[i % nsets]
        hip.call("atst_gemm_nt_bf16", hip.ptr(A), hip.ptr(B), M, N, K, K, K, epi, hip.ptr(out), N, hip.ptr(C2), hip.ptr(bias),
                 None, None, 256, None, None, None, None, None, hip.stream())
    for i in range(nsets): call(i)
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    n = 24; e0.record()
    for i in range(n): call(i)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print(f"{label:10s} N={N} K={K} sets={nsets:2d}: {ms*1e3:7.1f} us  {2.0*M*N*K/ms/1e9:6.1f} TF/s")
for nsets in (1, 12):
    run(1536, 384, hip.EPI_BIAS_GELU, nsets, "fc1+gelu")
    run(1152, 384, hip.EPI_BF16, nsets, "qkv")
