#!/usr/bin/env python3
"""Where the host thread spends the enqueue time of an ATST-Frame step (cProfile over 20 steps).  usage: python tools/debug/frame_host_profile.py [frame|clip6]"""
import cProfile, os, pstats, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
wl = sys.argv[1] if len(sys.argv) > 1 else "frame"
eng, step, _ = bench.build_job(wl, "small", "bf16", False, 256, 1, 0, torch.device("cuda:0"), 40, False, True)
for k in range(8): step(k)
torch.cuda.synchronize()
pr = cProfile.Profile(); t0 = time.perf_counter(); pr.enable()
for k in range(8, 28): step(k)
pr.disable(); t_host = time.perf_counter() - t0
torch.cuda.synchronize(); t_all = time.perf_counter() - t0
print(f"{wl}: host enqueue {t_host / 20 * 1e3:.2f} ms/step, wall {t_all / 20 * 1e3:.2f} ms/step")
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
