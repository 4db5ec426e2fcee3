#!/usr/bin/env python3
"""Per-collective times and overlap of the gradient all-reduce with the backward pass, from a rocprofv3 --kernel-trace CSV of
    ATST_FORCE_COLLECTIVES=1 python3 bench.py --steps N --warmup W --no-cpu-baseline --no-profile        (one rank through RCCL)
usage: python tools/comm_trace.py <trace-dir-or-csv> [out.txt]
For every step (delimited by adamw_ema_kernel): the RCCL kernels in launch order with their durations, how much of each ran
underneath compute kernels (interval intersection on the device timeline), and the exposed gap = start of the optimizer kernel
minus the end of the last backward kernel.  On ONE rank a collective moves no data over xGMI: the durations are the launch /
kernel floor of each collective and the overlap structure, not link time (SCALE_rNN.json has the 8-GPU curve)."""
import csv, glob, os, re, sys
src = sys.argv[1]
files = [src] if os.path.isfile(src) else glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
is_comm = lambda n: "nccl" in n.lower() or "rccl" in n.lower()
steps, cur = [], []
for r in rows:
    cur.append(r)
    if "adamw_ema_kernel" in r[2]:
        steps.append(cur); cur = []
steps = steps[2:]                                   # warm-up
out = []
def overlap(a0, a1, ivs):
    t = 0
    for b0, b1 in ivs:
        lo, hi = max(a0, b0), min(a1, b1)
        if hi > lo: t += hi - lo
    return t
tot_comm = tot_ov = tot_gap = 0.0
sample = None
for st in steps:
    comp = [(s, e) for s, e, n in st if not is_comm(n)]
    comm = [(s, e, n) for s, e, n in st if is_comm(n)]
    adam = [(s, e) for s, e, n in st if "adamw_ema_kernel" in n][0]
    last_bwd_end = max(e for s, e in comp if e <= adam[0])
    gap = (adam[0] - last_bwd_end) * 1e-3
    rec = []
    for s, e, n in comm:
        ov = overlap(s, e, comp)
        rec.append(((e - s) * 1e-3, ov * 1e-3, (s - st[0][0]) * 1e-3))
        tot_comm += (e - s) * 1e-3; tot_ov += ov * 1e-3
    tot_gap += gap
    if sample is None: sample = (rec, gap, (adam[1] - st[0][0]) * 1e-3)
n = len(steps)
out.append(f"# {n} steps; RCCL kernels per step: {len(sample[0])}; step length {sample[2]:.0f} us")
out.append(f"RCCL kernel time per step {tot_comm / n:8.1f} us, of which underneath compute kernels {tot_ov / n:8.1f} us ({100 * tot_ov / max(tot_comm, 1e-9):.1f} %)")
out.append(f"exposed gap (end of the last backward kernel -> start of adamw_ema_kernel) {tot_gap / n:8.1f} us per step")
out.append("one step, collectives in launch order:   start(us)   duration(us)   under compute(us)")
for d, ov, t0 in sample[0]:
    out.append(f"                                       {t0:10.0f} {d:12.1f} {ov:16.1f}")
txt = "\n".join(out)
print(txt)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(txt + "\n")
