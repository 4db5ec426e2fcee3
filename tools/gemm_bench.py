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
def nt_ln(K, label):
    """residual GEMM with the fused LayerNorm epilogue (what the encoder forward launches for proj / fc2)"""
    A = torch.randn(M, K, device=dev).bfloat16(); B = (torch.randn(384, K, device=dev) * 0.05).bfloat16()
    bias = torch.randn(384, device=dev); resid = torch.randn(M, 384, device=dev); scale = torch.ones(M // 256, device=dev)
    x = torch.empty(M, 384, device=dev); h = torch.empty(M, 384, device=dev, dtype=torch.bfloat16)
    g, b = torch.ones(384, device=dev), torch.zeros(384, device=dev); mean, rstd = torch.empty(M, device=dev), torch.empty(M, device=dev)
    f = lambda: hip.call("atst_gemm_nt_resid_ln_bf16", hip.ptr(A), hip.ptr(B), M, K, hip.ptr(bias), hip.ptr(resid), hip.ptr(scale), 256, hip.ptr(x),
                         hip.ptr(g), hip.ptr(b), hip.ptr(h), hip.ptr(mean), hip.ptr(rstd), hip.stream())
    ms = t_ms(f); fl = 2.0 * M * 384 * K; by = 2.0 * K * (M + 384) + 10.0 * M * 384
    print(f"  nt {label:22s} N=  384 K={K:5d}  {ms*1e3:8.1f} us  {fl/ms/1e9:7.1f} TF/s  {by/ms/1e9:6.2f} TB/s")
def nt_lnbwd(K, label):
    """dgrad GEMM with the LayerNorm backward as its epilogue, and the unfused pair (dgrad GEMM + ln_bwd kernel) it replaces"""
    dY = torch.randn(M, K, device=dev).bfloat16(); Wt = (torch.randn(384, K, device=dev) * 0.05).bfloat16()
    x = torch.randn(M, 384, device=dev); mean = x.mean(1).contiguous(); rstd = torch.rsqrt(x.var(1, unbiased=False) + 1e-6).contiguous()
    gamma = torch.ones(384, device=dev); dres = torch.randn(M, 384, device=dev); dx = torch.empty(M, 384, device=dev)
    g = torch.empty(M, 384, device=dev, dtype=torch.bfloat16); scale = torch.ones(M // 256, device=dev)
    dg, db, du = (torch.zeros(384, device=dev) for _ in range(3)); dh = torch.empty(M, 384, device=dev, dtype=torch.bfloat16)
    f = lambda: hip.call("atst_gemm_nt_lnbwd_bf16", hip.ptr(dY), hip.ptr(Wt), M, K, hip.ptr(x), hip.ptr(mean), hip.ptr(rstd), hip.ptr(gamma), hip.ptr(dres),
                         hip.ptr(dx), hip.ptr(g), hip.ptr(scale), 256, hip.ptr(dg), hip.ptr(db), hip.ptr(du), hip.stream())
    def unfused():
        hip.call("atst_gemm_nt_bf16", hip.ptr(dY), hip.ptr(Wt), M, 384, K, K, K, hip.EPI_BF16, hip.ptr(dh), 384, None, None, None, None, 1, None, None, None, None, None, hip.stream())
        hip.call("atst_layernorm_bwd", hip.ptr(dh), hip.ptr(x), hip.ptr(mean), hip.ptr(rstd), hip.ptr(gamma), hip.ptr(dres), hip.ptr(dx), hip.ptr(g), hip.ptr(scale), 256,
                 hip.ptr(dg), hip.ptr(db), hip.ptr(du), M, 384, hip.stream())
    ms, ms2 = t_ms(f), t_ms(unfused); by = 2.0 * K * (M + 384) + 14.0 * M * 384
    print(f"  nt {label:22s} N=  384 K={K:5d}  {ms*1e3:8.1f} us  {by/ms/1e9:6.2f} TB/s   (unfused GEMM + ln_bwd: {ms2*1e3:8.1f} us)")
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
nt_ln(384, "proj fwd (+resid+LN)")
nt_ln(1536, "fc2 fwd (+resid+LN)")
nt(1536, 384, hip.EPI_DGELU, "fc2 dgrad (+dgelu)")
nt_lnbwd(1536, "fc1 dgrad (+LN bwd)")
nt_lnbwd(1152, "qkv dgrad (+LN bwd)")
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
