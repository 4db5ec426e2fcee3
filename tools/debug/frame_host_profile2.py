#!/usr/bin/env python3
"""Pure host cost of enqueueing one step (the queue is drained before every step, so no back-pressure from the GPU shows up as host time):
python tools/debug/frame_host_profile2.py [frame|clip6|clip2] [small|base] [bf16|fp8]"""
import cProfile, os, pstats, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
wl = sys.argv[1] if len(sys.argv) > 1 else "frame"
arch = sys.argv[2] if len(sys.argv) > 2 else "small"
dt = sys.argv[3] if len(sys.argv) > 3 else "bf16"
eng, step, _ = bench.build_job(wl, arch, dt, False, 256, 1, 0, torch.device("cuda:0"), 40, False, True)
for k in range(8): step(k)
torch.cuda.synchronize()
pr = cProfile.Profile(); ts = []
for k in range(8, 28):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); pr.enable()
    step(k)
    pr.disable(); ts.append(time.perf_counter() - t0)
torch.cuda.synchronize()
ts.sort()
print(f"{wl} {arch} {dt}: host enqueue per step: median {ts[len(ts) // 2] * 1e3:.2f} ms, min {ts[0] * 1e3:.2f}, max {ts[-1] * 1e3:.2f}")
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
