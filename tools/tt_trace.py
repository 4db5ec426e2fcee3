#!/usr/bin/env python3
"""Stage timeline of one block of the two-team persistent GEMM (build with -DATST_TT_TRACE=<block + 1>): s_memtime stamps per wave and stage in rounds 2 .. 5.
usage (GPU box): ATST_LIB_TAG=tttrace python tools/tt_trace.py [N K epi]     stamps: 0 stage start, 1 vmcnt wait done, 2 barrier passed, 3 stage work done"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audiossl_amd import hip
lib = hip.load()
lib.atst_tune_gemm_variant(2001)
M = int(os.environ.get("M", 131072))
N, K = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1152, 384)
epi = int(sys.argv[3]) if len(sys.argv) > 3 else hip.EPI_BF16
A = torch.randn(M, K, device="cuda").bfloat16(); B = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16); C2 = torch.empty_like(out) if epi == hip.EPI_BIAS_GELU else None
bias = torch.randn(N, device="cuda")
dbg = torch.zeros(8 * 4 * 48 * 4, dtype=torch.int64, device="cuda")
for _ in range(3):
    dbg.zero_()
    hip.call("atst_gemm_nt_bf16", hip.ptr(A), hip.ptr(B), M, N, K, K, K, epi, hip.ptr(out), N, hip.ptr(C2), hip.ptr(bias), None, None, 256, None, None, None, None,
             hip.ptr(dbg.view(torch.float32)), hip.stream())
torch.cuda.synchronize()
t = dbg.cpu().view(8, 4, 48, 4).numpy().astype(np.int64)
nk = min(K // 64, 48)
t0 = t[t > 0].min()
print(f"N={N} K={K} epi={epi} nk={nk}: cycles; rounds 2..5; role of team 0 = ML in even rounds")
for r in range(4):
    rr = r + 2
    print(f"-- round {rr}: ML team {rr & 1}")
    for w in range(8):
        x = t[w, r, :nk]
        if (x == 0).any():
            print(f"  wave {w}: incomplete"); continue
        role = "ML" if (w >> 2) == (rr & 1) else "EP"
        wait = x[:, 1] - x[:, 0]; bar = x[:, 2] - x[:, 1]; work = x[:, 3] - x[:, 2]
        per = (x[nk - 1, 3] - x[0, 0]) / nk
        print(f"  wave {w} {role}: stage period {per:6.0f} | vmcnt wait {wait.mean():6.0f} (max {wait.max():5d}) | barrier {bar.mean():6.0f} (max {bar.max():5d}) | work {work.mean():6.0f} (max {work.max():5d})"
              f" | start {x[0, 0] - t0:7d}")
w0 = t[0, 0, :nk]; w4 = t[4, 0, :nk]
print("per stage, round 2: wave 0 [wait, barrier, work] | wave 4 [wait, barrier, work]")
for s in range(nk):
    print(f"  s={s:2d}  {w0[s,1]-w0[s,0]:5d} {w0[s,2]-w0[s,1]:5d} {w0[s,3]-w0[s,2]:5d}   |  {w4[s,1]-w4[s,0]:5d} {w4[s,2]-w4[s,1]:5d} {w4[s,3]-w4[s,2]:5d}")
