#!/usr/bin/env python3
"""Why is the grouped weight-gradient launch slower inside the step (~650 us at M = 131072) than alone (~390 us)?
Times atst_gemm_tn_group_bf16 on the block's four problems with (a) compact random operands, (b) operands scattered over a
multi-GB arena like the activation tape, (c) a fresh dW target per call (cold atomics), (d) zero operands (power / DVFS)."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audiossl_amd import hip
hip.load(); dev = "cuda"
M = int(os.environ.get("M", 131072)); Cd = 384
shapes = [(4 * Cd, Cd), (Cd, 4 * Cd), (3 * Cd, Cd), (Cd, Cd)]
def t_us(fn, n=12):
    for _ in range(2): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
def make(alloc, fill):
    ops = []
    for N, K in shapes:
        dY = alloc(M * N, torch.bfloat16).view(M, N); X = alloc(M * K, torch.bfloat16).view(M, K)
        fill(dY); fill(X); ops.append((dY, X))
    return ops
def run(ops, dws, label):
    items = (hip.Wgrad * 4)()
    k = [0]
    def call():
        dw = dws[k[0] % len(dws)]; k[0] += 1
        for i, ((N, K), (dY, X)) in enumerate(zip(shapes, ops)):
            items[i] = hip.Wgrad(hip.ptr(dY), hip.ptr(X), hip.ptr(dw[i]), M, N, K, N, K, K)
        hip.call("atst_gemm_tn_group_bf16", C.cast(items, C.c_void_p), 4, hip.stream())
    print(f"{label:70s} {t_us(call):8.1f} us")
rand = lambda t: t.copy_(torch.randn(t.shape, device=dev).to(t.dtype))
small = lambda t: t.copy_((torch.randn(t.shape, device=dev) * 1e-4).to(t.dtype))
zero = lambda t: t.zero_()
plain = lambda n, dt: torch.empty(n, dtype=dt, device=dev)
one_dw = [[torch.zeros(N, K, device=dev) for N, K in shapes]]
many_dw = [[torch.zeros(N, K, device=dev) for N, K in shapes] for _ in range(12)]
run(make(plain, rand), one_dw, "compact operands, N(0,1), one dW target")
run(make(plain, rand), many_dw, "compact operands, N(0,1), 12 dW targets in turn (cold atomics)")
run(make(plain, small), one_dw, "compact operands, N(0,1e-4) (gradient-sized values)")
run(make(plain, zero), one_dw, "compact operands, zeros")
arena = torch.empty(20 * 1024 ** 3, dtype=torch.uint8, device=dev); off = [0]
def scattered(n, dt):
    nb = n * torch.empty(0, dtype=dt).element_size()
    o = off[0]; off[0] += nb + 1800 * 1024 * 1024                       # one tensor per "layer" of a 1.8 GB tape
    off[0] %= (arena.numel() - nb - 1); off[0] -= off[0] % 256
    return arena[o:o + nb].view(dt)
run(make(scattered, rand), many_dw, "operands scattered over a 20 GB arena, N(0,1), 12 dW targets")
