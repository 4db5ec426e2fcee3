#!/usr/bin/env python3
"""Run one GEMM shape a few times (for rocprofv3 --pmc): python tools/one_gemm.py N K EPI"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audiossl_amd import hip
hip.load()
N, K, epi = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
M = 131072
A = torch.randn(M, K, device="cuda").bfloat16(); B = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
bias = torch.randn(N, device="cuda"); resid = torch.randn(M, N, device="cuda")
U = torch.randn(M, N, device="cuda").bfloat16()
out = torch.empty(M, N, device="cuda", dtype=torch.float32); C2 = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
scale = torch.ones(M // 256, device="cuda")
for _ in range(5):
    hip.call("atst_gemm_nt_bf16", hip.ptr(A), hip.ptr(B), M, N, K, K, K, epi, hip.ptr(out), N, hip.ptr(C2), hip.ptr(bias),
             hip.ptr(resid), hip.ptr(scale), 256, hip.ptr(U), None, None, None, None, hip.stream())
torch.cuda.synchronize()
