#!/usr/bin/env python3
"""Per-stage timeline of one block (blockIdx 100) of the grouped weight-gradient kernel (build with ATST_TRACE=1):
s_memtime at loop top / after the vmcnt wait / after barrier + next stage's LDS-DMA issue / after the stage's 36 MFMAs."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audiossl_amd import hip
lib = hip.load()
M, Cd = 131072, 384
dev = "cuda"
mk = lambda n: (torch.randn(M, n, device=dev) * 0.1).bfloat16()
du, h2, g, a, dqkv, h1, g2, o = mk(4 * Cd), mk(Cd), mk(Cd), mk(4 * Cd), mk(3 * Cd), mk(Cd), mk(Cd), mk(Cd)
dW = [torch.zeros(4 * Cd, Cd, device=dev), torch.zeros(Cd, 4 * Cd, device=dev), torch.zeros(3 * Cd, Cd, device=dev), torch.zeros(Cd, Cd, device=dev)]
items = (hip.Wgrad * 4)()
for it, (dy, x, w, N, K) in zip(items, [(du, h2, dW[0], 4 * Cd, Cd), (g, a, dW[1], Cd, 4 * Cd), (dqkv, h1, dW[2], 3 * Cd, Cd), (g2, o, dW[3], Cd, Cd)]):
    it.dY, it.X, it.dW, it.M, it.N, it.K, it.ldy, it.ldx, it.ldw = dy.data_ptr(), x.data_ptr(), w.data_ptr(), M, N, K, N, K, K
for _ in range(3):
    hip.check(lib.atst_gemm_tn_group_bf16(items, 4, hip.stream()))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); hip.check(lib.atst_gemm_tn_group_bf16(items, 4, hip.stream())); e1.record(); torch.cuda.synchronize()
print(f"grouped wgrad, M = {M}: {e0.elapsed_time(e1) * 1e3:.1f} us")
raw = C.CDLL(hip.LIB_PATH)
n = 8 * 260 * 4
buf = (C.c_ulonglong * n)()
assert raw.atst_debug_tn_trace(buf, n) == 0
t = np.frombuffer(buf, dtype=np.uint64).astype(np.int64).reshape(8, 260, 4)
nst = int((t[0, :, 0] > 0).sum()) - 1
sl = slice(5, min(nst, 200) - 5)
for w in (0, 3, 4, 7):
    x = t[w]
    per = (x[sl.stop, 0] - x[sl.start, 0]) / (sl.stop - sl.start)
    print(f" wave {w}: stages {nst}  period {per:7.0f} cycles | vmcnt wait {(x[sl, 1] - x[sl, 0]).mean():6.0f} | barrier + issue {(x[sl, 2] - x[sl, 1]).mean():6.0f} | "
          f"frag reads + 36 MFMAs {(x[sl, 3] - x[sl, 2]).mean():6.0f}")
print(" whole block:", int(t[0, 259, 0] - t[0, 0, 0]), "cycles for", nst, "stages")
