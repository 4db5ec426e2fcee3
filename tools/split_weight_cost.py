#!/usr/bin/env python3
"""What would two-term ([hi | lo]) bf16 weight operands cost at d = 384?  (measure it.)
W ~ hi + lo doubles the contraction of every forward and dgrad GEMM: C = [A | A] [hi | lo]^T.  A kernel that wraps the A operand's k-tile index would
stream the same LDS bytes and issue the same MFMAs as the plain kernel at 2 K with a materialised [A | A] -- only A's second pass would hit L2 instead of
HBM.  This script times the shipped kernels at K and at 2 K (materialised [A | A]: an UPPER bound by A's extra HBM bytes, 100-400 MB per launch) for the
eight GEMMs of an ATST-small block and converts the difference into step time with the launch counts of the clip2 / clip6 step.
Run on the GPU box: python tools/split_weight_cost.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audiossl_amd import hip
lib = hip.load(); dev = "cuda"; M = 131072


def timed(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def nt(N, K, epi):
    A = torch.randn(M, K, device=dev).bfloat16(); B = (torch.randn(N, K, device=dev) * 0.05).bfloat16(); bias = torch.randn(N, device=dev)
    resid = torch.randn(M, N, device=dev) if epi == hip.EPI_RESID else None; scale = torch.ones(M // 256, device=dev)
    U = torch.randn(M, N, device=dev).bfloat16() if epi == hip.EPI_DGELU else None
    out = torch.empty(M, N, device=dev, dtype=torch.float32 if epi == hip.EPI_RESID else torch.bfloat16)
    C2 = torch.empty(M, N, device=dev, dtype=torch.bfloat16) if epi == hip.EPI_BIAS_GELU else None
    return timed(lambda: hip.call("atst_gemm_nt_bf16", hip.ptr(A), hip.ptr(B), M, N, K, K, K, epi, hip.ptr(out), N, hip.ptr(C2), hip.ptr(bias), hip.ptr(resid),
                                  hip.ptr(scale) if resid is not None else None, 256, hip.ptr(U), None, None, None, None, hip.stream()))


def nt_ln(K):
    A = torch.randn(M, K, device=dev).bfloat16(); B = (torch.randn(384, K, device=dev) * 0.05).bfloat16()
    bias = torch.randn(384, device=dev); resid = torch.randn(M, 384, device=dev); scale = torch.ones(M // 256, device=dev)
    x = torch.empty(M, 384, device=dev); h = torch.empty(M, 384, device=dev, dtype=torch.bfloat16)
    g, b = torch.ones(384, device=dev), torch.zeros(384, device=dev); mean, rstd = torch.empty(M, device=dev), torch.empty(M, device=dev)
    return timed(lambda: hip.call("atst_gemm_nt_resid_ln_bf16", hip.ptr(A), hip.ptr(B), M, K, hip.ptr(bias), hip.ptr(resid), hip.ptr(scale), 256, hip.ptr(x),
                                  hip.ptr(g), hip.ptr(b), hip.ptr(h), hip.ptr(mean), hip.ptr(rstd), hip.stream()))


def nt_lnbwd(K):
    dY = torch.randn(M, K, device=dev).bfloat16(); Wt = (torch.randn(384, K, device=dev) * 0.05).bfloat16()
    x = torch.randn(M, 384, device=dev); mean = x.mean(1).contiguous(); rstd = torch.rsqrt(x.var(1, unbiased=False) + 1e-6).contiguous()
    gamma = torch.ones(384, device=dev); dres = torch.randn(M, 384, device=dev); dx = torch.empty(M, 384, device=dev)
    g = torch.empty(M, 384, device=dev, dtype=torch.bfloat16); scale = torch.ones(M // 256, device=dev)
    dg, db, du = (torch.zeros(384, device=dev) for _ in range(3))
    return timed(lambda: hip.call("atst_gemm_nt_lnbwd_bf16", hip.ptr(dY), hip.ptr(Wt), M, K, hip.ptr(x), hip.ptr(mean), hip.ptr(rstd), hip.ptr(gamma), hip.ptr(dres),
                                  hip.ptr(dx), hip.ptr(g), hip.ptr(scale), 256, hip.ptr(dg), hip.ptr(db), hip.ptr(du), hip.stream()))


# (label, timing function of K, K, launches per clip2 step at M = 131072: teacher forward 12 + student forward 12 ; backward 12)
rows = [("qkv forward", lambda K: nt(1152, K, hip.EPI_BF16), 384, 24), ("proj + residual + LN", nt_ln, 384, 24),
        ("fc1 + GELU", lambda K: nt(1536, K, hip.EPI_BIAS_GELU), 384, 24), ("fc2 + residual + LN", nt_ln, 1536, 24),
        ("fc2 dgrad + dGELU", lambda K: nt(1536, K, hip.EPI_DGELU), 384, 12), ("fc1 dgrad + LN backward", nt_lnbwd, 1536, 12),
        ("proj dgrad", lambda K: nt(384, K, hip.EPI_BF16), 384, 12), ("qkv dgrad + LN backward", nt_lnbwd, 1152, 12)]
extra = 0.0
print(f"M = {M}; us per launch at K | at 2 K | difference x launches per clip2 step")
for label, fn, K, n in rows:
    t1, t2 = fn(K), fn(2 * K)
    extra += (t2 - t1) * n
    print(f"  {label:26s} K = {K:4d}: {t1:7.1f} | {t2:7.1f} | +{t2 - t1:6.1f} us x {n} = {(t2 - t1) * n / 1e3:5.2f} ms", flush=True)
print(f"two-term weights: + {extra / 1e3:.2f} ms per clip2 step of global-view launches (the clip2 step is ~42 ms: + {extra / 1e3 / 42 * 100:.0f} %)")
