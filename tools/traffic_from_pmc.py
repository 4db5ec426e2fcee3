#!/usr/bin/env python3
"""HBM traffic per launch from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE cannot share a pass on gfx950).

    python tools/traffic_from_pmc.py <fetch_dir> <write_dir> <out.json> [workload] [HEAD the counters were taken at] [steps in the profiled run]

Units and corrections follow /opt/skills/guides/MI355X_MICROARCH.md "HBM": both counters are in KB; FETCH_SIZE counts
128-B fabric requests at 64 B, so it is doubled (checked here on the optimizer kernel, whose algorithmic read volume is
known exactly: see DESIGN.md "PMC findings").  Launches are grouped by the profile kind bench.py reports
(`gemm_nt_kernel<EPI:...>` covers both the 128x128 and the row-384 tiles of that epilogue)."""
import collections, csv, glob, json, re, sys

KIND = {"0": "gemm_nt_kernel<0:bf16>", "1": "gemm_nt_kernel<1:f32>", "2": "gemm_nt_kernel<2:bias_gelu>",
        "3": "gemm_nt_kernel<3:resid>", "4": "gemm_nt_kernel<4:dgelu>", "5": "gemm_nt_kernel<5:patch>", "6": "gemm_nt_kernel<6:lnbwd>"}


def kind_of(name: str) -> str:
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"gemm_nt(?:_row384|_w4)?_kernel<\(?(?:Epi\))?(\d)", name)
    if m:
        return KIND[m.group(1)]
    m = re.match(r"(gemm_tn)(?:_\w+)?_kernel", name)
    if m:
        return "gemm_tn_kernel"
    if name.startswith("attn_fwd"):
        return "attn_fwd_kernel"
    if name.startswith("attn_bwd") or name.startswith("attn_rowdot"):
        return "attn_bwd_dkv_kernel"
    m = re.search(r"(ln_fwd_kernel|ln_bwd_kernel|stft_mel_db_kernel|adamw_ema_kernel|byol_loss_kernel|patchify_kernel)", name)
    return m.group(1) if m else re.sub(r"[<(].*", "", name)


def load(d, counter):
    tot, cnt = collections.defaultdict(float), collections.Counter()
    for f in ([d] if d.endswith(".csv") else glob.glob(d + "/**/*counter_collection.csv", recursive=True)):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                k = kind_of(r["Kernel_Name"])
                tot[k] += float(r["Counter_Value"]); cnt[k] += 1
    return tot, cnt


def main():
    fd, wd, out = sys.argv[1:4]
    workload = sys.argv[4] if len(sys.argv) > 4 else "clip6"
    head = sys.argv[5] if len(sys.argv) > 5 else "unknown HEAD"
    steps = int(sys.argv[6]) if len(sys.argv) > 6 else 3               # round_measure.sh: --steps 2 --warmup 1
    ft, fc = load(fd, "FETCH_SIZE")
    wt, wc = load(wd, "WRITE_SIZE")
    rows = {}
    # whole-step HBM traffic: EVERY dispatch of the run (ATen copies included) divided by its steps
    step_bytes = (sum(ft.values()) * 1024.0 * 2.0 + sum(wt.values()) * 1024.0) / steps
    for k in sorted(set(ft) | set(wt)):
        if k.startswith("at::") or "elementwise" in k or "rccl" in k.lower():
            continue
        f = ft[k] / max(fc[k], 1) * 1024.0 * 2.0
        w = wt[k] / max(wc[k], 1) * 1024.0
        rows[k] = {"launches_fetch_pass": fc[k], "launches_write_pass": wc[k], "fetch_bytes_per_launch": round(f),
                   "write_bytes_per_launch": round(w), "traffic_bytes_per_launch": round(f + w)}
    json.dump({"workload": workload, "head": head, "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes",
               "corrections": "KB -> bytes; FETCH_SIZE x2 (gfx950 counts 128-B requests at 64 B)", "steps_profiled": steps,
               "step_traffic_bytes": round(step_bytes), "kernels": rows},
              open(out, "w"), indent=1)
    print(f"HBM traffic per step: {step_bytes / 1e9:.1f} GB over {steps} profiled steps")
    for k, v in sorted(rows.items(), key=lambda kv: -kv[1]["traffic_bytes_per_launch"] * max(kv[1]["launches_fetch_pass"], 1))[:14]:
        print(f"{k:36s} fetch {v['fetch_bytes_per_launch'] / 1e6:9.2f} MB  write {v['write_bytes_per_launch'] / 1e6:9.2f} MB  x{v['launches_fetch_pass']}")


if __name__ == "__main__":
    main()
