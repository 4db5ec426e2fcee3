#!/usr/bin/env python3
"""Coarse phase timeline (start / main loop / each epilogue part / stores drained) of one block of the 256x384 GEMM.
usage (GPU box): ATST_TRACE=301 python audiossl_amd/build.py && python tools/trace_epi.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audiossl_amd import hip
hip.load()
M = 131072
def run(N, K, epi, label):
    A = torch.randn(M, K, device="cuda").bfloat16(); B = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
    bias = torch.randn(N, device="cuda"); resid = torch.randn(M, N, device="cuda") if epi == hip.EPI_RESID else None
    out = torch.empty(M, N, device="cuda", dtype=torch.float32 if epi == hip.EPI_RESID else torch.bfloat16)
    C2 = torch.empty(M, N, device="cuda", dtype=torch.bfloat16) if epi == hip.EPI_BIAS_GELU else None
    scale = torch.ones(M // 256, device="cuda")
    dbg = torch.zeros(8 * 64 * 8, dtype=torch.int64, device="cuda")
    for _ in range(3):
        hip.call("atst_gemm_nt_bf16", hip.ptr(A), hip.ptr(B), M, N, K, K, K, epi, hip.ptr(out), N, hip.ptr(C2), hip.ptr(bias), hip.ptr(resid),
                 hip.ptr(scale) if resid is not None else None, 256, None, None, None, None, hip.ptr(dbg.view(torch.float32)), hip.stream())
    torch.cuda.synchronize()
    t = dbg.cpu().view(8, 64, 8)[:, :, 7].numpy()
    print(f"{label}: N={N} K={K}  (cycles; wave 0 / wave 7)")
    for w in (0, 7):
        x = t[w]
        print(f"  wave {w}: main loop {x[1]-x[0]:7d} | epilogue parts " + " ".join(f"{x[2+i]-x[1+i]:6d}" for i in range(8)) + f" | drain {x[10]-x[9]:6d} | total {x[10]-x[0]:7d}")
        if x[20]: print(f"           part 3: ds_write issue {x[20]-x[4]:5d} | loads issued, wait lgkm {x[21]-x[20]:5d} | barrier {x[22]-x[21]:5d} | read+compute+store {x[23]-x[22]:5d} | barrier {x[5]-x[23]:5d}")
run(1152, 384, hip.EPI_BF16, "qkv fwd")
run(1536, 384, hip.EPI_BIAS_GELU, "fc1+gelu")
run(384, 1536, hip.EPI_RESID, "fc2+resid")
run(384, 1536, hip.EPI_BF16, "fc1 dgrad")
