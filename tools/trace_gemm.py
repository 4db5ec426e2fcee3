#!/usr/bin/env python3
"""Phase timeline of one block of the row-384 GEMM main loop (build with ATST_TRACE=<block+1>): s_memtime stamps per wave.
usage (GPU box): ATST_TRACE=301 python audiossl_amd/build.py && VARIANT=321 python tools/trace_gemm.py [N K]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audiossl_amd import hip
hip.load()
if os.environ.get("VARIANT"): hip.load().atst_tune_gemm_variant(int(os.environ["VARIANT"]))
M = 131072
N, K = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (384, 1536)
A = torch.randn(M, K, device="cuda").bfloat16(); B = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
dbg = torch.zeros(8 * 64 * 8, dtype=torch.int64, device="cuda")
for _ in range(3):
    hip.call("atst_gemm_nt_bf16", hip.ptr(A), hip.ptr(B), M, N, K, K, K, hip.EPI_BF16, hip.ptr(out), N, None, None, None, None, 256, None, None, None, None,
             hip.ptr(dbg.view(torch.float32)), hip.stream())
torch.cuda.synchronize()
t = dbg.cpu().view(8, 64, 8).numpy()
nk = K // 32
pp = t[0, 1, 2] != 0
t0 = t[:, :nk, :][t[:, :nk, :] > 0].min()
print(f"N={N} K={K} nk={nk} ping-pong={bool(pp)}; cycles relative to first stamp; per wave: mean segment lengths over k-tiles 4..{nk-4}")
import numpy as np
sl = slice(4, nk - 4)
for w in range(8):
    x = t[w, :nk].astype(np.int64)
    if pp:
        comp = (x[sl, 1] - (x[sl, 0] if w < 4 else x[sl, 2])).mean()
        issue = (x[sl, 3] - (x[sl, 2] if w < 4 else x[sl, 0])).mean()
        wait = (x[sl, 4] - x[sl, 3]).mean()
        bar1 = ((x[sl, 2] - x[sl, 1]) if w < 4 else (x[sl, 2] - x[sl, 4])).mean()
        bar2 = ((x[sl, 5] - x[sl, 4]) if w < 4 else (x[sl, 5] - x[sl, 1])).mean()
        per = (x[nk - 4, 0] - x[4, 0]) / (nk - 8)
        print(f" wave {w}: k-tile period {per:7.0f}  compute {comp:6.0f}  dma-issue {issue:6.0f}  vmcnt-wait {wait:6.0f}  barrier-after-compute/-wait {bar1:6.0f}  other barrier {bar2:6.0f}")
    else:
        comp = (x[sl, 1] - x[sl, 0]).mean()
        per = (x[nk - 4, 0] - x[4, 0]) / (nk - 8)
        print(f" wave {w}: k-tile period {per:7.0f}  tile() {comp:6.0f}  wait+barrier {per - comp:6.0f}")
print("first k-tiles of wave 0 and wave 4 (start-of-tile stamps, relative):", (t[0, :6, 0] - t0).tolist(), (t[4, :6, 0] - t0).tolist())
print("main loop total (wave 0):", int(t[0, nk - 1, 1 if not pp else 5] - t[0, 0, 0]), "cycles")
