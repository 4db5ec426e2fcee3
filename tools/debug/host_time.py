#!/usr/bin/env python3
"""Host-side time of one training step (the time the Python thread needs to ENQUEUE it) next to the step's GPU time: if the two are close, a busy host
shows up as throughput.  usage: python tools/debug/host_time.py [clip6|clip2|frame] [arch] [dtype]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
wl = sys.argv[1] if len(sys.argv) > 1 else "frame"; arch = sys.argv[2] if len(sys.argv) > 2 else "small"; dtype = sys.argv[3] if len(sys.argv) > 3 else "bf16"
dev = torch.device("cuda:0")
eng, step, _ = bench.build_job(wl, arch, dtype, False, 256, 1, 0, dev, 40, False, True)
for k in range(8): step(k)
torch.cuda.synchronize()
host = []
t_all = time.perf_counter()
for k in range(8, 38):
    t0 = time.perf_counter(); step(k); host.append(time.perf_counter() - t0)
torch.cuda.synchronize()
wall = (time.perf_counter() - t_all) / 30
host.sort()
print(f"{wl} {arch} {dtype}: wall {wall * 1e3:.2f} ms/step ; host enqueue time per step: median {host[15] * 1e3:.2f} ms, min {host[0] * 1e3:.2f}, max {host[-1] * 1e3:.2f}")
