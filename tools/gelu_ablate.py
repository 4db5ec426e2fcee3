#!/usr/bin/env python3
"""A/B of the GELU epilogues across side-by-side builds (GPU box).  Build first, in the build container:
    ATST_LIB_TAG=gelu0 ATST_EXTRA_FLAGS=-DATST_GELU_MODE=0 python -m audiossl_amd.build   # round-3 forms (3e-7)
    ATST_LIB_TAG=gelu1 ATST_EXTRA_FLAGS=-DATST_GELU_MODE=1 python -m audiossl_amd.build   # no transcendental: the VALU ceiling
    python -m audiossl_amd.build                                                          # product (bf16-destination forms)
All libraries are loaded into ONE process and timed round-robin (medians), M = 131072: fc1 + GELU with / without the saved
pre-activation (student / teacher launches), fc2 dgrad + GELU' on the 128 x 128 tile and on the 256 x 384 tile."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audiossl_amd import hip
LIBDIR = os.path.join(os.path.dirname(hip.LIB_PATH))
tags = sys.argv[1:] or ["", "gelu0", "gelu1"]
libs = {t or "product": hip.load(os.path.join(LIBDIR, f"libatst_hip_{t}.so" if t else "libatst_hip.so")) for t in tags}
dev, M = "cuda", int(os.environ.get("M", 131072))
A = torch.randn(M, 384, device=dev).bfloat16(); W1 = (torch.randn(1536, 384, device=dev) * 0.05).bfloat16()
bias = torch.randn(1536, device=dev) * 0.1
u = torch.empty(M, 1536, device=dev, dtype=torch.bfloat16); a = torch.empty_like(u)
U = torch.randn(M, 1536, device=dev).bfloat16(); du = torch.empty_like(U); cs = torch.zeros(1536, device=dev)
st = hip.stream()
def fc1(lib, save_u):
    return lambda: hip.check(lib.atst_gemm_nt_bf16(hip.ptr(A), hip.ptr(W1), M, 1536, 384, 384, 384, hip.EPI_BIAS_GELU, hip.ptr(u) if save_u else None, 1536, hip.ptr(a),
                                                   hip.ptr(bias), None, None, 1, None, None, None, None, None, st))
def dgelu(lib, hook):
    def f():
        lib.atst_tune_gemm_variant(hook)
        hip.check(lib.atst_gemm_nt_bf16(hip.ptr(A), hip.ptr(W1), M, 1536, 384, 384, 384, hip.EPI_DGELU, hip.ptr(du), 1536, None, None, None, None, 1, hip.ptr(U),
                                        hip.ptr(cs), None, None, None, st))
    return f
cases = [("fc1+GELU (u saved)", lambda l: fc1(l, True)), ("fc1+GELU (teacher: no u)", lambda l: fc1(l, False)),
         ("fc2 dgrad+GELU' 128x128", lambda l: dgelu(l, 306)), ("fc2 dgrad+GELU' 256x384", lambda l: dgelu(l, 307))]
for name, mk in cases:
    fns = {k: mk(l) for k, l in libs.items()}
    for f in fns.values():
        for _ in range(3): f()
    times = {k: [] for k in fns}
    for rnd in range(7):
        for k, f in fns.items():
            torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): f()
            e1.record(); torch.cuda.synchronize()
            times[k].append(e0.elapsed_time(e1) / 10 * 1e3)
    print(f"{name:28s} " + "  ".join(f"{k}: {sorted(v)[len(v)//2]:7.1f} us" for k, v in times.items()), flush=True)
