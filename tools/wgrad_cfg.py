#!/usr/bin/env python3
"""Grouped weight-gradient launch (the four problems of one transformer block) under the wgrad schedule hooks 120 + c:
correctness against fp32 matmul of the same bf16 operands, then time at M = 131072 and M = 32768 (configurations take turns;
median / min / max over ROUNDS).  More schedules and the ablation switches (ATST_TN_ABL): apply
tools/experiments/wgrad_schedule_variants.patch to csrc/gemm.hip first (tools/wgrad_ablate.sh)."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audiossl_amd import hip
lib = hip.load(); dev = "cuda"
Cd = int(os.environ.get("D", 384))
shapes = [(4 * Cd, Cd), (Cd, 4 * Cd), (3 * Cd, Cd), (Cd, Cd)]
def t_us(fn, n=12):
    for _ in range(2): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
def group(ops, dws, M):
    items = (hip.Wgrad * 4)()
    for i, ((N, K), (dY, X)) in enumerate(zip(shapes, ops)):
        items[i] = hip.Wgrad(hip.ptr(dY), hip.ptr(X), hip.ptr(dws[i]), M, N, K, N, K, K)
    hip.call("atst_gemm_tn_group_bf16", C.cast(items, C.c_void_p), 4, hip.stream())
cfgs = [int(c) for c in os.environ.get("CFGS", "0,1,2").split(",")]
ROUNDS = int(os.environ.get("ROUNDS", 4))
for M in (131072, 32768):
    torch.manual_seed(0)
    ops = [(torch.randn(M, N, device=dev).bfloat16(), torch.randn(M, K, device=dev).bfloat16()) for N, K in shapes]
    dws = [torch.zeros(N, K, device=dev) for N, K in shapes]
    errs = {}
    if M == 32768:
        ref = [dY.float().t() @ X.float() for dY, X in ops]
        for c in cfgs:
            lib.atst_tune_gemm_variant(120 + c)
            for d in dws: d.zero_()
            group(ops, dws, M); torch.cuda.synchronize()
            errs[c] = "  rel err " + " ".join(f"{float((d - r).norm() / r.norm()):.1e}" for d, r in zip(dws, ref))
    # the configurations take turns (boxes drift by several per cent within a call): report the median and the minimum over the rounds
    times = {c: [] for c in cfgs}
    for _ in range(ROUNDS):
        for c in cfgs:
            lib.atst_tune_gemm_variant(120 + c)
            times[c].append(t_us(lambda: group(ops, dws, M), n=20))
    for c in cfgs:
        t = sorted(times[c])
        print(f"M={M:6d} cfg {c}: median {t[len(t) // 2]:8.1f} us  min {t[0]:8.1f}  max {t[-1]:8.1f}{errs.get(c, '')}", flush=True)
lib.atst_tune_gemm_variant(120)
