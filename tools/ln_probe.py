#!/usr/bin/env python3
"""LayerNorm forward / backward kernels stand-alone at M = 131072: time and effective HBM rate (C = 384, 768)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audiossl_amd import hip
hip.load(); dev = "cuda"
def t_us(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
M = 131072
for C in (384, 768):
    x = torch.randn(M, C, device=dev); gamma = torch.ones(C, device=dev); beta = torch.zeros(C, device=dev)
    y = torch.empty(M, C, device=dev, dtype=torch.bfloat16); mean = torch.empty(M, device=dev); rstd = torch.empty(M, device=dev)
    f = lambda: hip.call("atst_layernorm_fwd", hip.ptr(x), hip.ptr(gamma), hip.ptr(beta), hip.ptr(y), hip.ptr(mean), hip.ptr(rstd), M, C, hip.stream())
    t = t_us(f); print(f"C={C} ln_fwd  {t:7.1f} us  {6.0 * M * C / t / 1e6:5.2f} TB/s")
    dy = torch.randn(M, C, device=dev).bfloat16(); dres = torch.randn(M, C, device=dev); dx = torch.empty(M, C, device=dev)
    g = torch.empty(M, C, device=dev, dtype=torch.bfloat16); scale = torch.ones(M // 256, device=dev)
    dg, db, du = (torch.zeros(C, device=dev) for _ in range(3))
    f = lambda: hip.call("atst_layernorm_bwd", hip.ptr(dy), hip.ptr(x), hip.ptr(mean), hip.ptr(rstd), hip.ptr(gamma), hip.ptr(dres), hip.ptr(dx), hip.ptr(g), hip.ptr(scale), 256,
                         hip.ptr(dg), hip.ptr(db), hip.ptr(du), M, C, hip.stream())
    t = t_us(f); print(f"C={C} ln_bwd  {t:7.1f} us  {16.0 * M * C / t / 1e6:5.2f} TB/s")
