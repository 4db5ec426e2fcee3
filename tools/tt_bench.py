#!/usr/bin/env python3
"""Same-process A/B of the two-team persistent GEMM (csrc/gemm_tt.h, hook 2001) against the shipped kernels (hook 2000), round-robin per shape.
Run on the GPU box:  python tools/tt_bench.py [small|base|all]   (M from the environment, default 131072; LOCAL=1 adds the packed local-view M = 26624)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audiossl_amd import hip
lib = hip.load()
dev = "cuda"
which = sys.argv[1] if len(sys.argv) > 1 else "small"
HOOKS = [int(h) for h in os.environ.get("HOOKS", "2000,2001").split(",")]


def timed(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def shape(M, N, K, epi, label, save_u=True):
    A = torch.randn(M, K, device=dev).bfloat16(); B = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    bias = torch.randn(N, device=dev)
    U = torch.randn(M, N, device=dev).bfloat16() if epi == hip.EPI_DGELU else None
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    C2 = torch.empty(M, N, device=dev, dtype=torch.bfloat16) if epi == hip.EPI_BIAS_GELU else None
    f = lambda: hip.call("atst_gemm_nt_bf16", hip.ptr(A), hip.ptr(B), M, N, K, K, K, epi, hip.ptr(out) if save_u else None, N, hip.ptr(C2), hip.ptr(bias),
                         None, None, 256, hip.ptr(U), None, None, None, None, hip.stream())
    res = {h: [] for h in HOOKS}
    for rep in range(3):
        for h in HOOKS:
            lib.atst_tune_gemm_variant(h)
            res[h].append(timed(f))
    lib.atst_tune_gemm_variant(2000)
    fl = 2.0 * M * N * K
    by = 2.0 * K * (M + N) + 2.0 * M * N * ((2 if (epi == hip.EPI_BIAS_GELU and save_u) else 1) + (1 if epi == hip.EPI_DGELU else 0))
    line = f"  {label:26s} M={M:6d} N={N:5d} K={K:5d}"
    for h in HOOKS:
        med = sorted(res[h])[1]
        line += f" | {h}: {med:7.1f} us {fl / med / 1e6:6.0f} TF {by / med / 1e6:5.2f} TB/s"
    print(line, flush=True)


Ms = [int(os.environ.get("M", 131072))] + ([26624] if os.environ.get("LOCAL") else [])
for M in Ms:
    if which in ("small", "all"):
        shape(M, 1152, 384, hip.EPI_BF16, "qkv fwd")
        shape(M, 1536, 384, hip.EPI_BIAS_GELU, "fc1 + GELU (u, a)")
        shape(M, 1536, 384, hip.EPI_BIAS_GELU, "fc1 + GELU (a)", save_u=False)
        shape(M, 384, 384, hip.EPI_BF16, "proj dgrad")
        shape(M, 384, 1536, hip.EPI_BF16, "fc1 dgrad (plain)")
        shape(M, 1536, 384, hip.EPI_DGELU, "fc2 dgrad + dGELU")
    if which in ("base", "all"):
        shape(M, 2304, 768, hip.EPI_BF16, "base qkv fwd")
        shape(M, 3072, 768, hip.EPI_BIAS_GELU, "base fc1 + GELU (u, a)")
        shape(M, 3072, 768, hip.EPI_BIAS_GELU, "base fc1 + GELU (a)", save_u=False)
        shape(M, 768, 768, hip.EPI_BF16, "base proj dgrad")
        shape(M, 768, 3072, hip.EPI_BF16, "base fc1 dgrad (plain)")
        shape(M, 3072, 768, hip.EPI_DGELU, "base fc2 dgrad + dGELU")
