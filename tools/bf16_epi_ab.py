#!/usr/bin/env python3
"""Store-only bf16 GEMM epilogue, A/B in one process (run on the GPU box): hook 370 = fp32 staging through LDS, 32 rows per part, two block barriers per
part (rounds 1-3) ; hook 371 = transposed accumulators, bf16 wave-private staging, no block barrier (round 4).  Round-robin medians; also checks
that both produce the same bits.  usage: python tools/bf16_epi_ab.py [M]"""
import os, sys, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audiossl_amd import hip
lib = hip.load()
dev = "cuda"
M = int(sys.argv[1]) if len(sys.argv) > 1 else 131072


def run(N, K, bias, label, reps=7, inner=10):
    A = torch.randn(M, K, device=dev).bfloat16(); B = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    b = torch.randn(N, device=dev) if bias else None
    outs = [torch.empty(M, N, device=dev, dtype=torch.bfloat16) for _ in range(2)]
    def f(o):
        hip.call("atst_gemm_nt_bf16", hip.ptr(A), hip.ptr(B), M, N, K, K, K, hip.EPI_BF16, hip.ptr(o), N, None, hip.ptr(b),
                 None, None, 256, None, None, None, None, None, hip.stream())
    times = {0: [], 1: []}
    for rep in range(reps + 1):
        for v in (0, 1):
            lib.atst_tune_gemm_variant(370 + v)
            f(outs[v]); torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(inner): f(outs[v])
            e1.record(); torch.cuda.synchronize()
            if rep: times[v].append(e0.elapsed_time(e1) / inner * 1e3)
    same = torch.equal(outs[0], outs[1])
    t0, t1 = statistics.median(times[0]), statistics.median(times[1])
    by = 2.0 * K * (M + N) + 2.0 * M * N
    print(f"  {label:28s} N={N:5d} K={K:5d}  staged fp32 {t0:7.1f} us  transposed {t1:7.1f} us  ({100 * (t1 / t0 - 1):+5.1f} %)  "
          f"{by / t1 / 1e6:5.2f} TB/s  {2.0 * M * N * K / t1 / 1e6:6.0f} TF/s  bits equal: {same}")


print(f"M = {M}")
run(1152, 384, True, "qkv (small)")
run(384, 384, False, "proj dgrad (small)")
run(1536, 384, False, "N=1536 K=384")
run(384, 1536, False, "N=384 K=1536")
run(2304, 768, True, "qkv (base)")
run(768, 768, False, "proj dgrad (base)")
run(768, 3072, False, "fc1 dgrad shape (base)")
lib.atst_tune_gemm_variant(370)
