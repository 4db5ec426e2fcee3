#!/usr/bin/env python3
"""ATST-base (d = 768) GEMM shapes at M = 131072 under the tuning hooks, round-robin inside one process, with the library GEMM
(torch.matmul -> hipBLASLt, plain bf16 output) of the same shape as a reference point.  Run on the GPU box.
  VARIANTS="0;332;333"  ';'-separated hook sets (',' inside a set), 0 = defaults
"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audiossl_amd import hip
lib = hip.load()
dev = "cuda"
M = int(os.environ.get("M", 131072))
D = int(os.environ.get("D", 768))
VARS = [[int(x) for x in v.split(",") if x] for v in os.environ.get("VARIANTS", "390;391").split(";")]
RESET = [-1, 106, 111, 301, 304, 308, 330, 350, 361, 370, 381, 393]          # the shipped defaults of every hook family


def set_variant(vs):
    for r in RESET: lib.atst_tune_gemm_variant(r)
    for v in vs:
        if v: lib.atst_tune_gemm_variant(v)


def t_us(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def mk_nt(N, K, epi, save_u=True):
    A = torch.randn(M, K, device=dev).bfloat16(); B = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    bias = torch.randn(N, device=dev); resid = torch.randn(M, N, device=dev) if epi == hip.EPI_RESID else None
    U = torch.randn(M, N, device=dev).bfloat16() if epi == hip.EPI_DGELU else None
    out = torch.empty(M, N, device=dev, dtype=torch.float32 if epi in (hip.EPI_F32, hip.EPI_RESID) else torch.bfloat16)
    C2 = torch.empty(M, N, device=dev, dtype=torch.bfloat16) if epi == hip.EPI_BIAS_GELU else None
    colsum = torch.zeros(N, device=dev) if epi == hip.EPI_DGELU else None
    scale = torch.ones(M // 256, device=dev)
    keep = (A, B, bias, resid, U, out, C2, scale, colsum)
    c_ptr = hip.ptr(out) if (save_u or epi != hip.EPI_BIAS_GELU) else None
    f = lambda: hip.call("atst_gemm_nt_bf16", hip.ptr(A), hip.ptr(B), M, N, K, K, K, epi, c_ptr, N, hip.ptr(C2), hip.ptr(bias),
                         hip.ptr(resid), hip.ptr(scale) if resid is not None else None, 256, hip.ptr(U), None, None, None, hip.ptr(colsum), hip.stream())
    def ref():
        torch.matmul(A, B.t(), out=out if out.dtype == torch.bfloat16 else C2 if C2 is not None else keep_ref[0])
    keep_ref = [torch.empty(M, N, device=dev, dtype=torch.bfloat16)] if out.dtype != torch.bfloat16 else [None]
    return f, ref, keep


H = 4 * D
shapes = [("qkv fwd", 3 * D, D, hip.EPI_BF16, True), ("proj fwd +resid", D, D, hip.EPI_RESID, True), ("fc1 +gelu (u, a)", H, D, hip.EPI_BIAS_GELU, True),
          ("fc1 +gelu (a only)", H, D, hip.EPI_BIAS_GELU, False), ("fc2 fwd +resid", D, H, hip.EPI_RESID, True), ("fc2 dgrad +dgelu", H, D, hip.EPI_DGELU, True),
          ("fc1 dgrad", D, H, hip.EPI_BF16, True), ("proj dgrad", D, D, hip.EPI_BF16, True), ("qkv dgrad", D, 3 * D, hip.EPI_BF16, True)]
only = os.environ.get("ONLY")
print(f"M={M} D={D}  variants={VARS}")
for name, N, K, epi, save_u in shapes:
    if only and only not in name: continue
    f, ref, keep = mk_nt(N, K, epi, save_u)
    fl = 2.0 * M * N * K
    res = []
    for rep in range(2):
        for vs in VARS:
            set_variant(vs)
            try:
                res.append((tuple(vs), t_us(f)))
            except hip.HipError as e:
                res.append((tuple(vs), float("nan")))
    set_variant([])
    tr = t_us(ref) if os.environ.get("BLAS", "1") == "1" else float("nan")
    line = f"  {name:20s} N={N:5d} K={K:5d} "
    for vs in VARS:
        ts = [t for v, t in res if v == tuple(vs)]
        line += f"| {','.join(map(str, vs)):>8s}: {min(ts):7.1f} us {fl / min(ts) / 1e6:6.0f} TF "
    line += f"| blas(bf16 out) {tr:7.1f} us {fl / tr / 1e6:6.0f} TF"
    print(line, flush=True)
    del f, ref, keep
    torch.cuda.empty_cache()
