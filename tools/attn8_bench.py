#!/usr/bin/env python3
"""NP = 256 attention at the ATST-base geometry (S = 512 sequences, 12 heads): the e4m3 outputs written by the kernels themselves next to the bf16
kernels + separate quantisation pass they replace (run on the GPU box)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audiossl_amd import hip
hip.load(); dev = "cuda"; S, H, NP = int(os.environ.get("S", 512)), 12, 256
C = 64 * H
def t_us(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
qkv = torch.randn(S * NP, 3 * C, device=dev).bfloat16(); vt = torch.full((S,), 251, dtype=torch.int32, device=dev)
o = torch.empty(S * NP, C, dtype=torch.bfloat16, device=dev); o8 = torch.empty(S * NP, C, dtype=torch.uint8, device=dev)
lse = torch.empty(S, H, NP, device=dev); sc = torch.tensor([8.0], device=dev); site = torch.zeros(hip.AMAX_SITE_STRIDE, device=dev)
sat = torch.zeros(1, dtype=torch.int32, device=dev); st = hip.stream()
f0 = t_us(lambda: hip.call("atst_attention_fwd", hip.ptr(qkv), hip.ptr(vt), hip.ptr(o), hip.ptr(lse), S, H, NP, st))
q = t_us(lambda: hip.call("atst_quant_fp8_dyn_bf16", hip.ptr(o), o.numel(), hip.ptr(sc), hip.ptr(o8), hip.ptr(site), st))
f1 = t_us(lambda: hip.call("atst_attention_fwd_fp8", hip.ptr(qkv), hip.ptr(vt), hip.ptr(o), hip.ptr(o8), hip.ptr(sc), hip.ptr(site), hip.ptr(sat), hip.ptr(lse), S, H, NP, st))
f2 = t_us(lambda: hip.call("atst_attention_fwd_fp8", hip.ptr(qkv), hip.ptr(vt), None, hip.ptr(o8), hip.ptr(sc), hip.ptr(site), hip.ptr(sat), hip.ptr(lse), S, H, NP, st))
print(f"forward : bf16 {f0:.1f} us + quantisation pass {q:.1f} us = {f0 + q:.1f} ; bf16 + e4m3 from the kernel {f1:.1f} ; e4m3 only {f2:.1f}")
d_o = torch.randn(S * NP, C, device=dev).bfloat16(); dqkv = torch.empty_like(qkv); d8 = torch.empty(S * NP, 3 * C, dtype=torch.uint8, device=dev)
scr = torch.empty(S, H, NP, device=dev); gs = torch.tensor([4096.0], device=dev)
b0 = t_us(lambda: hip.call("atst_attention_bwd", hip.ptr(qkv), hip.ptr(vt), hip.ptr(o), hip.ptr(lse), hip.ptr(d_o), hip.ptr(dqkv), hip.ptr(scr), S, H, NP, st))
qb = t_us(lambda: hip.call("atst_quant_fp8_dyn_bf16", hip.ptr(dqkv), dqkv.numel(), hip.ptr(gs), hip.ptr(d8), hip.ptr(site), st))
b2 = t_us(lambda: hip.call("atst_attention_bwd_fp8", hip.ptr(qkv), hip.ptr(vt), hip.ptr(o), hip.ptr(lse), hip.ptr(d_o), hip.ptr(d8), hip.ptr(gs), hip.ptr(site), hip.ptr(scr), S, H, NP, st))
print(f"backward: bf16 {b0:.1f} us + quantisation pass {qb:.1f} us = {b0 + qb:.1f} ; e4m3 only {b2:.1f}")
