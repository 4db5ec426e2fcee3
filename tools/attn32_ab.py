#!/usr/bin/env python3
"""NP = 32 attention backward at (about) the local-view geometry of one step (1024 sequences, 6 heads; padded rows here, packed in the step):
two kernels (hook 408) against the fused one-wave-per-pair kernel (409).  GPU box."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audiossl_amd import hip
lib = hip.load()
S, H, NP = 1024, 6, 32
C = H * 64
qkv = torch.randn(S * NP, 3 * C, device="cuda").bfloat16()
vt = torch.full((S,), 26, dtype=torch.int32, device="cuda")
o = torch.empty(S * NP, C, dtype=torch.bfloat16, device="cuda"); lse = torch.empty(S, H, NP, device="cuda")
hip.call("atst_attention_fwd", hip.ptr(qkv), hip.ptr(vt), hip.ptr(o), hip.ptr(lse), S, H, NP, hip.stream())
d_o = torch.randn(S * NP, C, device="cuda").bfloat16(); dqkv = torch.empty_like(qkv)
f = lambda: hip.call("atst_attention_bwd", hip.ptr(qkv), hip.ptr(vt), hip.ptr(o), hip.ptr(lse), hip.ptr(d_o), hip.ptr(dqkv), None, S, H, NP, hip.stream())
for v in (408, 409, 408, 409):
    lib.atst_tune_gemm_variant(v)
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): f()
    e1.record(); torch.cuda.synchronize()
    print(f"hook {v} ({'two kernels' if v == 408 else 'fused'}): {e0.elapsed_time(e1) / 50 * 1e3:7.1f} us")
