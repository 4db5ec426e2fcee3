#!/usr/bin/env python3
"""Attention forward / backward time at the bench geometry (S = 512 sequences of 256 tokens, 6 heads), median of 5 x 10 launches."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audiossl_amd import hip
lib = hip.load(); dev = "cuda"
S, H, NP = int(os.environ.get("S", 512)), int(os.environ.get("H", 6)), 256
C = 64 * H
qkv = torch.randn(S * NP, 3 * C, device=dev).bfloat16(); valid = torch.full((S,), 251, dtype=torch.int32, device=dev)
o = torch.empty(S * NP, C, device=dev, dtype=torch.bfloat16); lse = torch.empty(S, H, NP, device=dev)
d_o = torch.randn(S * NP, C, device=dev).bfloat16(); dqkv = torch.empty_like(qkv); scr = torch.empty(S, H, NP, device=dev)
def med(fn):
    ts = []
    for _ in range(5):
        torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 100)
    return sorted(ts)[2]
fwd = lambda: hip.call("atst_attention_fwd", hip.ptr(qkv), hip.ptr(valid), hip.ptr(o), hip.ptr(lse), S, H, NP, hip.stream())
bwd = lambda: hip.call("atst_attention_bwd", hip.ptr(qkv), hip.ptr(valid), hip.ptr(o), hip.ptr(lse), hip.ptr(d_o), hip.ptr(dqkv), hip.ptr(scr), S, H, NP, hip.stream())
fwd(); bwd()
print(f"attention fwd {med(fwd):7.1f} us   bwd (row-dot + merged kernel) {med(bwd):7.1f} us", flush=True)
if os.environ.get("AB"):                         # hooks 407 / 406: LDS-transposed full-line dK / dV stores (default) / row-per-lane stores, taking turns
    for rnd in range(3):
        for v, name in ((7, "full-line"), (6, "row-per-lane")):
            lib.atst_tune_gemm_variant(400 + v)
            print(f"  bwd dK/dV stores {name}: {med(bwd):7.1f} us", flush=True)
    lib.atst_tune_gemm_variant(407)
