#!/usr/bin/env python3
"""dGELU GEMM (fc2 dgrad + GELU' + fc1 bias gradient) on the 128x128 and the 256x384 tile, small and base geometry, with and without
the column sums; correctness of the column sums against torch."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audiossl_amd import hip
lib = hip.load(); dev = "cuda"
def t_us(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
for d in (384, 768):
    M, N, K = 131072, 4 * d, d
    A = (torch.randn(M, K, device=dev) * 0.1).bfloat16(); B = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    U = torch.randn(M, N, device=dev).bfloat16(); out = torch.empty(M, N, device=dev, dtype=torch.bfloat16); cs = torch.zeros(N, device=dev)
    def call(colsum):
        hip.call("atst_gemm_nt_bf16", hip.ptr(A), hip.ptr(B), M, N, K, K, K, hip.EPI_DGELU, hip.ptr(out), N, None, None, None, None, 256, hip.ptr(U), None, None, None,
                 hip.ptr(cs) if colsum else None, hip.stream())
    for v, name in ((306, "128x128"), (307, "256x384")):
        lib.atst_tune_gemm_variant(v)
        cs.zero_(); call(True); torch.cuda.synchronize()
        ref = out.float().sum(0)
        err = float((cs - ref).norm() / ref.norm())
        a = sorted(t_us(lambda: call(False)) for _ in range(3))[1]; b = sorted(t_us(lambda: call(True)) for _ in range(3))[1]
        print(f"d={d} {name}: no colsum {a:7.1f} us   with colsum {b:7.1f} us   colsum vs sum of the bf16 output: {err:.1e}", flush=True)
lib.atst_tune_gemm_variant(308)
