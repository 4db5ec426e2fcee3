#!/usr/bin/env python3
"""A few launches of the NP=256 attention kernels at bench geometry (for rocprofv3 --pmc runs)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audiossl_amd import hip
hip.load()
if os.environ.get("ATTNV"): hip.load().atst_tune_gemm_variant(400 + int(os.environ["ATTNV"]))
dev = "cuda"
S, H, NP = int(os.environ.get("S", 512)), 6, 256
qkv = torch.randn(S * NP, 1152, device=dev).bfloat16(); valid = torch.full((S,), 251, dtype=torch.int32, device=dev)
o = torch.empty(S * NP, 384, device=dev, dtype=torch.bfloat16); lse = torch.empty(S, H, NP, device=dev)
d_o = torch.randn(S * NP, 384, device=dev).bfloat16(); dqkv = torch.empty_like(qkv); scr = torch.empty(S, H, NP, device=dev)
for _ in range(int(os.environ.get("N", 5))):
    hip.call("atst_attention_fwd", hip.ptr(qkv), hip.ptr(valid), hip.ptr(o), hip.ptr(lse), S, H, NP, hip.stream())
    if os.environ.get("BWD", "1") == "1":
        hip.call("atst_attention_bwd", hip.ptr(qkv), hip.ptr(valid), hip.ptr(o), hip.ptr(lse), hip.ptr(d_o), hip.ptr(dqkv), hip.ptr(scr), S, H, NP, hip.stream())
torch.cuda.synchronize()
