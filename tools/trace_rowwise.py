#!/usr/bin/env python3
"""Phase timeline of one block's row-wise epilogue (residual + LayerNorm, LayerNorm backward).
The stamps are not part of the product kernel: apply tools/experiments/rowwise_epilogue_trace.patch (patch -p0 from the repo root; it
also restores the start-up skew hook 100000 + c), build with ATST_EXTRA_FLAGS=-DATST_EPI_TRACE=301 (block 300 = second round), run
on the GPU box.  Round-3 result: profiles/r03_trace_rowwise.txt."""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audiossl_amd import hip
lib = C.CDLL(hip.LIB_PATH)
hip.load()
dev = "cuda"
M = int(os.environ.get("M", 131072))
buf = torch.zeros(8 * 128, dtype=torch.int64, device=dev)
lib.atst_debug_epi_trace.argtypes = [C.c_void_p]
def show(label):
    torch.cuda.synchronize()
    t = buf.cpu().view(8, 128).tolist()
    for w in (0, 7):
        r = t[w]
        print(f"{label} wave {w}: main-loop end -> all stores acknowledged {r[49] - r[0]} cycles")
        print("   part:  staging  wait-in  barrier   pair0   pair1  end-bar")
        prev = r[0]
        for part in range(8):
            st = r[1 + 6 * part: 1 + 6 * part + 6]
            d = [st[0] - prev] + [st[i] - st[i - 1] for i in range(1, 6)]
            prev = st[5]
            print(f"   {part}:   " + " ".join(f"{v:7d}" for v in d))
def run_ln(K):
    buf.zero_()
    A = torch.randn(M, K, device=dev).bfloat16(); B = (torch.randn(384, K, device=dev) * 0.05).bfloat16()
    bias = torch.randn(384, device=dev); resid = torch.randn(M, 384, device=dev); scale = torch.ones(M // 256, device=dev)
    x = torch.empty(M, 384, device=dev); h = torch.empty(M, 384, device=dev, dtype=torch.bfloat16)
    g, b = torch.ones(384, device=dev), torch.zeros(384, device=dev); mean, rstd = torch.empty(M, device=dev), torch.empty(M, device=dev)
    for _ in range(3):
        hip.call("atst_gemm_nt_resid_ln_bf16", hip.ptr(A), hip.ptr(B), M, K, hip.ptr(bias), hip.ptr(resid), hip.ptr(scale), 256, hip.ptr(x),
                 hip.ptr(g), hip.ptr(b), hip.ptr(h), hip.ptr(mean), hip.ptr(rstd), hip.stream())
    show(f"resid+LN K={K}")
def run_lnb(K):
    buf.zero_()
    dY = torch.randn(M, K, device=dev).bfloat16(); Wt = (torch.randn(384, K, device=dev) * 0.05).bfloat16()
    x = torch.randn(M, 384, device=dev); mean = x.mean(1).contiguous(); rstd = torch.rsqrt(x.var(1, unbiased=False) + 1e-6).contiguous()
    gamma = torch.ones(384, device=dev); dres = torch.randn(M, 384, device=dev); dx = torch.empty(M, 384, device=dev)
    g = torch.empty(M, 384, device=dev, dtype=torch.bfloat16); scale = torch.ones(M // 256, device=dev)
    dg, db, du = (torch.zeros(384, device=dev) for _ in range(3))
    for _ in range(3):
        hip.call("atst_gemm_nt_lnbwd_bf16", hip.ptr(dY), hip.ptr(Wt), M, K, hip.ptr(x), hip.ptr(mean), hip.ptr(rstd), hip.ptr(gamma), hip.ptr(dres),
                 hip.ptr(dx), hip.ptr(g), hip.ptr(scale), 256, hip.ptr(dg), hip.ptr(db), hip.ptr(du), hip.stream())
    show(f"LN-bwd K={K}")
lib.atst_debug_epi_trace(C.c_void_p(buf.data_ptr()))
run_ln(384); run_ln(1536); run_lnb(1536)
lib.atst_debug_epi_trace(None)
