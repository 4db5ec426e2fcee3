#!/usr/bin/env python3
"""Per-shape timing of the encoder GEMMs / attention / LN at the bench geometry (run on the GPU box)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audiossl_amd import hip
hip.load()
dev = "cuda"
if os.environ.get("VARIANT"): hip.load().atst_tune_gemm_variant(int(os.environ["VARIANT"]))
if os.environ.get("ATTNV"): hip.load().atst_tune_gemm_variant(400 + int(os.environ["ATTNV"]))
if os.environ.get("TNV"): hip.load().atst_tune_gemm_variant(100 + int(os.environ["TNV"]))
M = int(os.environ.get("M", 131072))
def t_ms(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
def nt(N, K, epi, label):
    A = torch.randn(M, K, device=dev).bfloat16(); B = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    bias = torch.randn(N, device=dev); resid = torch.randn(M, N, device=dev) if epi == hip.EPI_RESID else None
    U = torch.randn(M, N, device=dev).bfloat16() if epi == hip.EPI_DGELU else None
    out = torch.empty(M, N, device=dev, dtype=torch.float32 if epi in (hip.EPI_F32, hip.EPI_RESID) else torch.bfloat16)
    C2 = torch.empty(M, N, device=dev, dtype=torch.bfloat16) if epi == hip.EPI_BIAS_GELU else None
    scale = torch.ones(M // 256, device=dev)
    f = lambda: hip.call("atst_gemm_nt_bf16", hip.ptr(A), hip.ptr(B), M, N, K, K, K, epi, hip.ptr(out), N, hip.ptr(C2), hip.ptr(bias),
                         hip.ptr(resid), hip.ptr(scale) if resid is not None else None, int(os.environ.get('RPS', 256)), hip.ptr(U), None, None, None, None, hip.stream())
    ms = t_ms(f); fl = 2.0 * M * N * K
    print(f"  nt {label:22s} N={N:5d} K={K:5d}  {ms*1e3:8.1f} us  {fl/ms/1e9:7.1f} TF/s")
def tn(N, K, label):
    dY = torch.randn(M, N, device=dev).bfloat16(); X = torch.randn(M, K, device=dev).bfloat16(); dW = torch.zeros(N, K, device=dev)
    f = lambda: hip.call("atst_gemm_tn_bf16", hip.ptr(dY), hip.ptr(X), M, N, K, N, K, hip.ptr(dW), K, int(os.environ.get("SPLIT", 0)), hip.stream())
    ms = t_ms(f); fl = 2.0 * M * N * K
    print(f"  tn {label:22s} N={N:5d} K={K:5d}  {ms*1e3:8.1f} us  {fl/ms/1e9:7.1f} TF/s")
def attn():
    S, H, NP = M // 256, 6, 256
    qkv = torch.randn(S * NP, 1152, device=dev).bfloat16(); valid = torch.full((S,), 251, dtype=torch.int32, device=dev)
    o = torch.empty(S * NP, 384, device=dev, dtype=torch.bfloat16); lse = torch.empty(S, H, NP, device=dev)
    d_o = torch.randn(S * NP, 384, device=dev).bfloat16(); dqkv = torch.empty_like(qkv); scr = torch.empty(S, H, NP, device=dev)
    ms = t_ms(lambda: hip.call("atst_attention_fwd", hip.ptr(qkv), hip.ptr(valid), hip.ptr(o), hip.ptr(lse), S, H, NP, hip.stream()))
    print(f"  attn fwd  {ms*1e3:8.1f} us  {4.0*S*H*251*251*64/ms/1e9:7.1f} TF/s (algorithmic)")
    ms = t_ms(lambda: hip.call("atst_attention_bwd", hip.ptr(qkv), hip.ptr(valid), hip.ptr(o), hip.ptr(lse), hip.ptr(d_o), hip.ptr(dqkv), hip.ptr(scr) if os.environ.get("ATTNB", "1") == "1" else None, S, H, NP, hip.stream()))
    print(f"  attn bwd  {ms*1e3:8.1f} us  {8.0*S*H*251*251*64/ms/1e9:7.1f} TF/s (algorithmic, 2x fwd)")
print(f"M={M}")
nt(1152, 384, hip.EPI_BF16, "qkv fwd")
nt(384, 384, hip.EPI_RESID, "proj fwd (+resid)")
nt(1536, 384, hip.EPI_BIAS_GELU, "fc1 fwd (+gelu)")
nt(384, 1536, hip.EPI_RESID, "fc2 fwd (+resid)")
nt(1536, 384, hip.EPI_DGELU, "fc2 dgrad (+dgelu)")
nt(384, 1536, hip.EPI_BF16, "fc1 dgrad")
nt(384, 384, hip.EPI_BF16, "proj dgrad")
nt(384, 1152, hip.EPI_BF16, "qkv dgrad")
tn(384, 1536, "fc2 wgrad"); tn(1536, 384, "fc1 wgrad"); tn(384, 384, "proj wgrad"); tn(1152, 384, "qkv wgrad")
attn()
if os.environ.get("EXTRA"):
    for spec in os.environ["EXTRA"].split(","):
        m_, n_, k_ = (int(v) for v in spec.split("x"))
        M = m_
        nt(n_, k_, hip.EPI_BF16, f"extra M={m_}")
