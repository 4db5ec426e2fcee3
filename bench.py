#!/usr/bin/env python3
"""Headline benchmark: ATST-small pre-training clips/s on synthetic 10 s @ 16 kHz waveforms (BASELINE.json).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload clip6|clip2|frame] [--batch B]

One timed step = mel front end of every view (HIP) -> teacher fwd -> student fwd -> loss -> student bwd ->
(RCCL all-reduce of student grads at N>1) -> fused HF-AdamW + EMA teacher.  Inputs (12 s waveform buffers) are resident
in HBM before the timed region.  Prints ONE JSON line on rank 0 (contract in the task statement): metric / value /
roofline (dominant kernel, live HIP-event timing from inside libatst_hip.so) / cpu_baseline (oracle on host cores).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0      # MI355X dense bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_FP8_TFLOPS = 5000.0       # dense e4m3 MFMA peak (same table): what a kernel on v_mfma_scale_f32_32x32x64_f8f6f4 is priced against
PEAK_HBM_GBS = 8000.0


def flops_per_clip(workload: str, d=384, depth=12, patch_k=256) -> float:
    """Algorithmic GEMM FLOPs per clip (SURVEY.md 8(d) / BASELINE.md 2): fwd = 1x, bwd = 2x, padding not counted."""
    def enc(N, P):
        return depth * (24 * N * d * d + 4 * N * N * d) + 2 * P * patch_k * d
    proj, pred = 2 * (d * 4096 + 4096 * 256), 2 * (256 * 4096 + 4096 * 256)
    if workload == "clip2":
        return 2 * enc(251, 250) + 3 * 2 * enc(251, 250) + 2 * proj + 3 * 2 * (proj + pred)
    if workload == "clip6":
        g, l = enc(251, 250), enc(26, 25)
        return 2 * g + 3 * (2 * g + 4 * l) + 2 * proj + 3 * 6 * (proj + pred)
    if workload == "frame":
        g, rows = enc(250, 250), 163
        return 2 * g + 3 * 2 * g + rows * (2 * proj + 3 * 2 * (proj + pred))
    raise ValueError(workload)


def cpu_baseline(seconds_budget=25.0):
    """The reference's algorithm on the host cores: the CPU oracle (plain fp32 torch restatement pinned to the
    reference by tests/test_oracle_golden.py) on BASELINE.json configs[0]: B=2 clips x 2 views, 10 s, mel front end +
    teacher fwd + student fwd/bwd + HF-AdamW + EMA.  Bounded sample (SURVEY 8(d) protocol): 2 warm-up + 5 timed steps per
    setting, median; ~20 s of CPU work on the GPU box's 64 cores."""
    from oracle import atst_oracle as O
    ncpu = os.cpu_count() or 1
    out = {}
    for label, threads in (("all", min(ncpu, 64)), ("one", 1)):
        torch.set_num_threads(threads)
        W = O.recipe_weights("small", seed=0)
        leaves = {k: v.requires_grad_(True) for k, v in W.items()
                  if k.startswith("student.") and v.dtype == torch.float32 and "running" not in k}
        st = {k: (torch.zeros_like(v), torch.zeros_like(v)) for k, v in leaves.items()}
        reg, _ = O.param_groups([(k[len("student."):], tuple(v.shape)) for k, v in leaves.items()])
        wave = O.recipe_wave(2, 192000, seed=1234)
        times, t_end = [], time.time() + seconds_budget * (0.4 if label == "all" else 1.2)
        for step in range(1, 8):
            t0 = time.time()
            mels = [O.log_mel(wave[:, o:o + 160000]) for o in (1000, 20000)]
            lens = [torch.full((2,), 1001)] * 2
            loss, _, _ = O.atst_forward(W, mels, lens, "small", 2)
            for v in leaves.values():
                v.grad = None
            loss.backward()
            with torch.no_grad():
                for k, v in leaves.items():
                    if v.grad is not None:
                        O.hf_adamw_step(v, v.grad, st[k][0], st[k][1], step, 5e-4, 0.04 if k[len("student."):] in reg else 0.0)
            for v in leaves.values():
                v.requires_grad_(False)
            O.ema_update(W, 0.99)
            for v in leaves.values():
                v.requires_grad_(True)
            if step > 2:
                times.append(time.time() - t0)
            if time.time() > t_end and len(times) >= 3:             # slow host: never fewer than 3 timed steps
                break
        times.sort()
        out[label] = (2.0 / times[len(times) // 2], threads, len(times))
    torch.set_num_threads(min(ncpu, 64))
    v, c, n = out["all"]
    return {"value": round(v, 3), "unit": "clips/s", "cores": c, "kind": "port",
            "sample": f"oracle (fp32 torch CPU restatement), configs[0]: B=2 clips x 2 views x 10 s, mel + teacher fwd + "
                      f"student fwd/bwd + HF-AdamW + EMA, median of {n} steps after 2 warm-ups",
            "one_thread_value": round(out["one"][0], 3), "one_thread_note": "reference as shipped pins OMP/MKL to 1 thread"}


def launch_ranks(n: int) -> int:
    """`python bench.py --gpus N` outside a launcher: start N ranks (one per GPU) as fresh child processes through
    torch.distributed.run and relay rank 0's JSON line.  This parent has made no HIP / torch.cuda call, and it never
    exec()s: the children are ordinary subprocesses (ref: Trainer(devices=nproc, strategy="ddp"), methods/atst/train.py:18-32)."""
    import socket
    import subprocess
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    for l in p.stdout.splitlines():
        if not l.startswith("{"):
            print(l, file=sys.stderr)
    if p.returncode != 0 or not lines:
        print(f"bench.py: the {n}-rank launch failed (exit code {p.returncode})", file=sys.stderr)
        return p.returncode or 1
    print(lines[-1], flush=True)
    return 0


def build_job(workload, arch, dtype, hires, B, world, rank, dev, total, precise=False, overlap=False):
    """Engine + resident synthetic input + the step closure of one workload (the headline, or one of the `also` passes).
    One step = mel front end of every view -> teacher fwd -> student fwd -> loss -> student bwd -> (all-reduce) -> fused HF-AdamW + EMA."""
    from audiossl_amd.engine import AtstEngine
    from audiossl_amd.frontend import LogMelFrontend
    from audiossl_amd.utils.common import cosine_scheduler_step
    frame = workload == "frame"
    ncrops = 6 if workload == "clip6" else 2
    sr, n_mels, patch = (32000, 128, (128, 8)) if hires else (16000, 64, (64, 4))
    clip_len, buf_len = 10 * sr, 12 * sr
    eng = AtstEngine(arch, frame=frame, ncrops=ncrops, fp8=dtype == "fp8", patch_h=patch[0], patch_w=patch[1], precise=precise)
    eng.init_weights(seed=0)
    eng.broadcast_parameters()                                       # DDP init: every rank takes rank 0's replica (no-op at world 1)
    eng.overlap_teacher = overlap
    fe = LogMelFrontend(1024 if not frame else 640, sr=sr, n_mels=n_mels)
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    buf = torch.clamp(0.1 * torch.randn(B, buf_len, device=dev, generator=g), -1.0, 1.0)      # 12 s, resident in HBM
    lr_tab = cosine_scheduler_step(5e-4 * world * B / 256, 1e-6, 39100, 1300)
    wd_tab = cosine_scheduler_step(0.04, 0.4, 39100, 0)
    ema_tab = cosine_scheduler_step(0.99, 1, 39100, 0)
    cpu_gen = torch.Generator().manual_seed(99 + rank)
    offs = torch.randint(0, buf_len - clip_len, (total, 6), generator=cpu_gen).tolist()
    mask_sets = None
    if frame:
        import numpy as np
        rs = np.random.RandomState(1234 + rank)
        from audiossl_amd.methods.atstframe.random_mask import block_mask
        # host-side masks, one draw per clip per step, as the DataLoader would deliver them (the transform draws them
        # with numpy on the workers: methods/atstframe/transform.py:84-101); generated before the timed region
        mask_sets = [torch.from_numpy(np.stack([block_mask(250, 0.65, 5, rng=rs) for _ in range(B)])) for _ in range(total)]

    # every view group gets ONE [V * B, 1, n_mels, T] buffer: the front end reads the waveform slices in place (row stride) and
    # writes each view into its rows, so the engine sees a ready concatenated group (no slice copies, no torch.cat)
    n_glob = 1 if frame else 2
    g_buf = torch.empty(n_glob * B, 1, n_mels, 1 + clip_len // 160, device=dev)
    l_buf = torch.empty(4 * B, 1, n_mels, 1 + (clip_len // 10) // 160, device=dev) if ncrops == 6 else None
    Tg, Tl = 1 + clip_len // 160, 1 + (clip_len // 10) // 160

    def step(k):
        o = offs[k]
        masks = None
        if frame:
            mel = fe(buf[:, o[0]:o[0] + clip_len], out=g_buf)
            mels, lens = [mel, mel], [torch.full((B,), Tg)] * 2
            masks = [mask_sets[k], mask_sets[k]]              # ONE mask shared by both views (transform.py:99)
        else:
            mels = [fe(buf[:, o[v]:o[v] + clip_len], out=g_buf[v * B:(v + 1) * B]) for v in range(2)]
            lens = [torch.full((B,), Tg)] * 2
            if ncrops == 6:                                  # 4 local views of 1 s -> 101 frames -> 25 patches + CLS
                mels += [fe(buf[:, o[2 + v]:o[2 + v] + clip_len // 10], out=l_buf[v * B:(v + 1) * B]) for v in range(4)]
                lens += [torch.full((B,), Tl)] * 4
        loss, _, _ = eng.forward(mels, lens, masks)
        eng.backward()
        eng.allreduce_grads()
        eng.optimizer_step(float(lr_tab[k]), float(wd_tab[k]), float(ema_tab[k + 1]))
        return loss

    return eng, step, dict(frame=frame, ncrops=ncrops, patch=patch, khz="32kHz" if hires else "16kHz")


WORKLOAD_NAMES = {"clip6": "ATST-small clip-level, 2 global (10 s) + 4 local (1 s) views", "clip2": "ATST-small clip-level, 2 views (10 s)",
                  "frame": "ATST-Frame small, masked frame objective (10 s)"}
DTYPE_NAMES = {"bf16": "bf16", "fp8": "fp8 (e4m3 operands in all 12 GEMMs of a block: forward, dgrads, weight gradients; delayed scaling) + bf16"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100, help="timed steps (100 x ~60 ms: the timed region dominates the run)")
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="clip6", choices=["clip6", "clip2", "frame"])
    ap.add_argument("--batch", type=int, default=256, help="clips per GPU per step")
    ap.add_argument("--arch", default="small", choices=["small", "base"],
                    help="small = the headline model (BASELINE configs[1..3]); base (d = 768, 12 heads) is an extra data point")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp8"],
                    help="fp8: the four Linear layers of every block run their forward AND their dgrad and weight-gradient GEMMs on OCP e4m3 operands "
                         "(MX-scaled MFMA, delayed scaling; the attention backward writes dqkv as e4m3); saved activations the attention needs stay bf16 "
                         "(BASELINE.json configs[4]: use with --arch base)")
    ap.add_argument("--precise", action="store_true",
                    help="parity mode (AtstEngine(precise=True)): fp32 activations / gradients, every Linear as a split-bf16 MFMA GEMM (~2^-16); "
                         "the mode that meets north_star's 1e-3 gradient tolerance -- reported next to the bf16 number, never as the headline")
    ap.add_argument("--hires", action="store_true",
                    help="BASELINE.json configs[4] input geometry: 10 s @ 32 kHz, 128 mel bands, one patch row of 128 x 8 (the reference's "
                         "sr / n_mels / patch_h / patch_w parameters) -> 2001 frames, 250 patches of 1024 values; use with --arch base --dtype fp8")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--also", dest="also", action="store_true", default=None,
                    help="after the headline's timed region, short passes of the other BASELINE.json configurations (ATST-Frame, 2-view, ATST-base bf16 / fp8 / "
                         "fp8 at 32 kHz) attached as `also` in the same JSON line; default: on for the default headline at one GPU")
    ap.add_argument("--no-also", dest="also", action="store_false")
    ap.add_argument("--also-steps", type=int, default=10)
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--profile-stride", type=int, default=17,
                    help="HIP events around one launch in n of each kernel kind, a hashed subset of the launch indices (1 = all: costs ~6 %% of the step; 7: ~0.6 %%; "
                         "17: ~0.3 %% -- events fence the overlap of consecutive kernels and of the two streams)")
    ap.add_argument("--overlap", action="store_true", help="run the teacher pass on a second HIP stream")
    ap.add_argument("--backend", default=os.environ.get("ATST_DIST_BACKEND", "nccl"), choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (one rank per GPU); gloo = test transport, lets several ranks share one GPU")
    args = ap.parse_args()

    if args.also is None:                                           # the driver's default command gets them; explicit workloads / A-B runs do not
        args.also = (args.workload, args.arch, args.dtype, args.hires, args.batch, args.precise) == ("clip6", "small", "bf16", False, 256, False) \
            and not args.no_cpu_baseline and not args.no_profile
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus))                           # parent: never touches the GPU, relays rank 0's line
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks", file=sys.stderr)
        sys.exit(2)
    ndev = torch.cuda.device_count()
    if args.backend == "nccl" and world > ndev:
        print(f"bench.py: {world} ranks need {world} GPUs, this node has {ndev}", file=sys.stderr)
        sys.exit(2)
    local = local % max(ndev, 1)                                     # gloo test transport: ranks may share a device
    torch.cuda.set_device(local)
    if world > 1 or os.environ.get("ATST_FORCE_COLLECTIVES") == "1":
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", local)
    torch.manual_seed(4321 + rank)                                   # DropPath draws (device generator): reproducible runs

    from audiossl_amd import hip
    lib = hip.load()

    B = args.batch
    if args.precise:
        if args.dtype != "bf16":
            ap.error("--precise is the fp32 parity mode; it excludes --dtype fp8")
        args.no_profile = True                                      # the parity-mode kernels are not instrumented
    total = args.warmup + args.steps + 2
    eng, step, info = build_job(args.workload, args.arch, args.dtype, args.hires, B, world, rank, dev, total, args.precise, args.overlap)
    frame, ncrops, patch = info["frame"], info["ncrops"], info["patch"]

    def sync():
        if dist.is_initialized():
            dist.barrier(device_ids=[local]) if args.backend == "nccl" else dist.barrier()
        torch.cuda.synchronize()

    for k in range(args.warmup):
        step(k)
    sync()
    if not args.no_profile:
        lib.atst_profile_enable(args.profile_stride)
    t0 = time.perf_counter()
    for k in range(args.warmup, args.warmup + args.steps):
        loss = step(k)
    sync()
    dt = time.perf_counter() - t0
    lib.atst_profile_enable(0)
    if os.environ.get("ATST_BENCH_MEM"):        # soak runs: allocator state after the timed region (stderr; not part of the JSON contract)
        print(f"[bench] rank {rank}: max allocated {torch.cuda.max_memory_allocated() / 2**30:.2f} GiB, reserved {torch.cuda.memory_reserved() / 2**30:.2f} GiB",
              file=sys.stderr, flush=True)
    tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
    if dist.is_initialized():
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax)
    loss_val = float(loss)

    roof, kernels, step_hbm = None, [], None
    ridge = PEAK_BF16_TFLOPS * 1e12 / (PEAK_HBM_GBS * 1e9)              # FLOP per HBM byte where the two roofs meet
    ridge8 = PEAK_FP8_TFLOPS * 1e12 / (PEAK_HBM_GBS * 1e9)              # ... for a kernel on e4m3 operands

    def is_fp8_kernel(name):
        """In an fp8 pass the encoder GEMMs (every nt epilogue but the fp32-output head Linears and the patch embed) and the weight gradients of a
        block run on e4m3 operands (v_mfma_scale_f32_32x32x64_f8f6f4); a few small launches of the same class (heads, patch weight gradient) stay bf16."""
        return (name.startswith("gemm_nt_kernel<") and not name.startswith(("gemm_nt_kernel<1", "gemm_nt_kernel<5"))) or name == "gemm_tn_kernel"

    def kernel_table(fp8=False):
        """Per-kernel-class table from the in-library HIP-event records collected since the last call (clears them).  fp8: the pass ran the fp8 engine --
        its e4m3 kernels meet the HBM roof at 5000 / 8 = 625 FLOP per byte, not at 312."""
        nk = lib.atst_profile_kinds()
        ms, work, byts, cnt = (C.c_double * nk)(), (C.c_double * nk)(), (C.c_double * nk)(), (C.c_longlong * nk)()
        hip.check(lib.atst_profile_collect(ms, work, byts, cnt), "atst_profile_collect")
        tab = []
        for i in range(nk):
            if cnt[i]:
                name = lib.atst_profile_name(i).decode()
                has_flops = "gemm" in name or "attn" in name
                # a kernel whose algorithmic intensity is below the ridge is bounded by HBM, whatever unit executes it
                mfma = has_flops and work[i] / byts[i] >= (ridge8 if (fp8 and is_fp8_kernel(name)) else ridge)
                secs = ms[i] * 1e-3
                rec = {"kernel": name, "launches": int(cnt[i]), "avg_us": round(ms[i] / cnt[i] * 1e3, 2),
                       "total_ms": round(ms[i], 3), "bound": "mfma" if mfma else "hbm",
                       "achieved": round(work[i] / secs / 1e12 if mfma else byts[i] / secs / 1e9, 2),
                       "unit": "TFLOP/s" if mfma else "GB/s", "bytes_per_launch": round(byts[i] / cnt[i])}
                if has_flops:
                    rec["tflops"] = round(work[i] / secs / 1e12, 2)
                    rec["flop_per_byte"] = round(work[i] / byts[i], 1)
                if name == "stft_mel_db_kernel":                        # `achieved` stays its algorithmic HBM bytes / time, but HBM is not what bounds it
                    rec["bound"] = "lds"
                    rec["note"] = "LDS-bandwidth-bound (three in-place FFT passes through LDS, DESIGN.md section 3); its HBM traffic is ~0.03 ms per launch"
                tab.append(rec)
        tab.sort(key=lambda r: -r["total_ms"])
        return tab

    exclusive = None
    if not args.no_profile:
        kernels = kernel_table(args.dtype == "fp8")                        # live: the timed region, as it ran
        # With the second stream on (clip6: local-view groups beside the teacher pass / the global-view backward) a launch's event time includes the
        # time it shares the chip with the other chain -- the step is faster, every overlapped launch looks slower.  A kernel's distance from its
        # roof is a property of the kernel alone: the same classes are therefore timed once more, OUTSIDE the timed region, with the second stream
        # off (same process, same data, same instrumentation) and reported next to the live numbers.
        if getattr(eng, "overlap_local_teacher", False) and ncrops == 6 and not frame:
            eng.overlap_local_teacher = False
            lib.atst_profile_enable(args.profile_stride)
            n_ex = min(args.steps, 10)
            for k in range(args.warmup, args.warmup + n_ex):
                step(k)
            sync()
            lib.atst_profile_enable(0)
            exclusive = {r["kernel"]: r for r in kernel_table(args.dtype == "fp8")}
            eng.overlap_local_teacher = True
        if kernels:
            # the dominant class: largest share of kernel time when nothing overlaps (= kernels[0] when there is no second stream)
            # (live and exclusive tables are each a hashed subset of the launches: a class missing from the live one falls back to its largest)
            d = kernels[0] if not exclusive else next((r for r in kernels if r["kernel"] == max(exclusive.values(), key=lambda x: x["total_ms"])["kernel"]), kernels[0])
            f8k = args.dtype == "fp8" and d["bound"] == "mfma" and is_fp8_kernel(d["kernel"])
            peak = (PEAK_FP8_TFLOPS if f8k else PEAK_BF16_TFLOPS) if d["bound"] == "mfma" else PEAK_HBM_GBS
            roof = {"bound": d["bound"], "kernel": d["kernel"], "achieved": d["achieved"], "peak": peak, "unit": d["unit"],
                    "frac": round(d["achieved"] / peak, 4), "traffic": None, "algorithmic_bytes_per_launch": d["bytes_per_launch"],
                    "avg_launch_us": d["avg_us"], "launches": d["launches"], "timed_one_launch_in": args.profile_stride,
                    "timed_every_nth_launch": args.profile_stride,      # the key's name before round 4 (kept as an alias)
                    "share_of_timed_kernel_ms": round(d["total_ms"] / sum(r["total_ms"] for r in kernels), 3)}
            if "tflops" in d:
                roof["tflops"], roof["flop_per_byte"] = d["tflops"], d["flop_per_byte"]
                roof["mfma_frac"] = round(d["tflops"] / PEAK_BF16_TFLOPS, 4)       # the same kernel priced against the dense bf16 MFMA peak
            if exclusive and d["kernel"] in exclusive:
                x = exclusive[d["kernel"]]
                roof["concurrency"] = ("two HIP streams: the local-view groups run beside the teacher pass (forward) and beside the global-view group "
                                       "(backward); achieved / frac / avg_launch_us above are LIVE event times of the timed region and include the time a "
                                       "launch shares the chip with the other chain -- `exclusive` is the same kernel class with the second stream off")
                roof["exclusive"] = {"achieved": x["achieved"], "frac": round(x["achieved"] / peak, 4), "unit": x["unit"], "avg_launch_us": x["avg_us"],
                                     "launches": x["launches"], "share_of_timed_kernel_ms": round(x["total_ms"] / sum(r["total_ms"] for r in exclusive.values()), 3),
                                     "how": f"{min(args.steps, 10)} extra steps after the timed region with ATST_OVERLAP_LT off, same HIP-event instrumentation"}
                if "tflops" in x:
                    roof["exclusive"]["mfma_frac"] = round(x["tflops"] / PEAK_BF16_TFLOPS, 4)
            # 10 s sequences are padded to 256 rows (251 / 250 real tokens); 1 s views are packed (26 rows for 26 tokens, unless
            # ATST_PACK=0): what the real tokens alone amount to
            loc = 26 if os.environ.get("ATST_PACK", "1") != "0" else 32
            real = {"clip2": 251 / 256, "frame": 250 / 256, "clip6": (3 * 2 * 251 + 4 * 26 * 1.0) / (3 * 2 * 256 + 4 * loc * 1.0)}[args.workload]
            roof["real_token_fraction"] = round(real, 4)
            roof["achieved_real_tokens_only"] = round(d["achieved"] * real, 2)
            # HBM bytes per launch from the committed PMC passes of this same command (rocprofv3 cannot collect
            # counters while the step is being timed); tools/round_measure.sh regenerates the file.
            tf = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", f"traffic_{args.workload}.json")
            if os.path.exists(tf) and args.batch == 256 and args.arch == "small":
                tj = json.load(open(tf))
                t = tj["kernels"].get(d["kernel"])
                src = (f"profiles/traffic_{args.workload}.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes"
                       f"; taken at {tj.get('head', 'an earlier HEAD')})")
                if t:
                    roof["traffic"] = t["traffic_bytes_per_launch"]
                    roof["traffic_source"] = src
                if tj.get("step_traffic_bytes"):                    # whole-step HBM traffic (every dispatch), priced against this run's step time
                    step_hbm = {"traffic_bytes_per_step": tj["step_traffic_bytes"], "source": src}
            # the kernel nearest the ridge of the two roofs (it reaches neither): priced against both
            near = [r for r in kernels if "tflops" in r and r["flop_per_byte"] >= 0.5 * ridge]
            if near:
                n0 = max(near, key=lambda r: r["total_ms"])
                roof["near_ridge_kernel"] = {"kernel": n0["kernel"], "flop_per_byte": n0["flop_per_byte"], "tflops": n0["tflops"],
                                             "mfma_frac": round(n0["tflops"] / PEAK_BF16_TFLOPS, 4),
                                             "hbm_GBs": round(n0["bytes_per_launch"] / (n0["avg_us"] * 1e-6) / 1e9, 1),
                                             "hbm_frac": round(n0["bytes_per_launch"] / (n0["avg_us"] * 1e-6) / 1e9 / PEAK_HBM_GBS, 4),
                                             "share_of_timed_kernel_ms": round(n0["total_ms"] / sum(r["total_ms"] for r in kernels), 3)}

    # ---- the other configurations of BASELINE.json, short passes OUTSIDE the timed region: configs[2] (ATST-Frame), the
    # 2-view recipe, and the single-GPU content of configs[4] (ATST-base, e4m3 GEMMs, 32 kHz / 128 mel / 128 x 8 patches).  New engines, freed
    # afterwards; same process (no exec, no re-launch); headline fields unchanged.
    also = []
    if args.also and world == 1 and not args.precise:
        headline = (args.workload, args.arch, args.dtype, args.hires)
        del eng, step
        import gc
        gc.collect()                                                # engine <-> pass reference cycles: without the collection every pass's workspace stayed allocated (258 GiB after eight passes)
        torch.cuda.empty_cache()
        for wl, arch, dtype, hires in (("frame", "small", "bf16", False), ("clip2", "small", "bf16", False), ("clip2", "base", "bf16", False),
                                       ("clip2", "base", "fp8", False), ("clip2", "base", "fp8", True),
                                       ("frame", "base", "bf16", False), ("frame", "base", "fp8", False),    # ATST-Frame base: the reference's train_base.sh recipe
                                       ("clip6", "small", "fp8", False), ("frame", "small", "fp8", False)):   # all-e4m3 + fp8_lean at d = 384 (round 6)
            if (wl, arch, dtype, hires) == headline:
                continue
            n_w, n_t = (8 if wl == "frame" else 3), args.also_steps       # ATST-Frame: its head batch is the masked rows of the step -- a few more steps until the row-buffer capacities have all been seen by the allocator
            eng2, step2, info2 = build_job(wl, arch, dtype, hires, B, world, rank, dev, n_w + n_t + 2)
            for k in range(n_w):
                step2(k)
            sync()
            lib.atst_profile_enable(5)
            t1 = time.perf_counter()
            for k in range(n_w, n_w + n_t):
                step2(k)
            sync()
            dt2 = time.perf_counter() - t1
            lib.atst_profile_enable(0)
            tab = kernel_table(dtype == "fp8")
            fpc2 = flops_per_clip(wl, d=768 if arch == "base" else 384, patch_k=info2["patch"][0] * info2["patch"][1])
            v2 = B * n_t / dt2
            rec = {"workload": WORKLOAD_NAMES[wl].replace("small", arch) + (", 10s@32kHz, 128 mel, 128 x 8 patches" if hires else ", 10s@16kHz"),
                   "arch": arch, "dtype": DTYPE_NAMES[dtype], "clips_per_gpu": B, "steps": n_t, "warmup": n_w, "value": round(v2, 2), "unit": "clips/s",
                   "ms_per_step": round(dt2 / n_t * 1e3, 3), "flops_per_clip_G": round(fpc2 / 1e9, 2),
                   "mfma_roofline_frac_step": round(v2 * fpc2 / 1e12 / PEAK_BF16_TFLOPS, 4), "mfma_roofline_frac_step_peak": "bf16 dense (2500 TFLOP/s)"}
            if dtype == "fp8":                                      # the same step priced against the peak of the dtype its GEMMs run in
                rec["mfma_roofline_frac_step_of_bf16_peak"] = rec["mfma_roofline_frac_step"]
                rec["mfma_roofline_frac_step_of_dtype_peak"] = round(v2 * fpc2 / 1e12 / PEAK_FP8_TFLOPS, 4)
            if tab:
                d2 = tab[0]
                f8k = dtype == "fp8" and d2["bound"] == "mfma" and is_fp8_kernel(d2["kernel"])
                pk = (PEAK_FP8_TFLOPS if f8k else PEAK_BF16_TFLOPS) if d2["bound"] == "mfma" else PEAK_HBM_GBS
                rec["dominant_kernel"] = {"kernel": d2["kernel"], "bound": d2["bound"], "achieved": d2["achieved"], "unit": d2["unit"], "peak": pk,
                                          "frac": round(d2["achieved"] / pk, 4), "avg_launch_us": d2["avg_us"],
                                          "share_of_timed_kernel_ms": round(d2["total_ms"] / sum(r["total_ms"] for r in tab), 3), "timed_one_launch_in": 5}
                if f8k:
                    rec["dominant_kernel"]["peak_is"] = "e4m3 dense MFMA"
                    rec["dominant_kernel"]["frac_of_bf16_peak"] = round(d2["achieved"] / PEAK_BF16_TFLOPS, 4)
                if "tflops" in d2:
                    rec["dominant_kernel"]["mfma_frac"] = round(d2["tflops"] / (PEAK_FP8_TFLOPS if f8k else PEAK_BF16_TFLOPS), 4)
            also.append(rec)
            del eng2, step2
            gc.collect()
            torch.cuda.empty_cache()
            if os.environ.get("ATST_BENCH_MEM"):
                print(f"[bench] after `also` {wl} {arch} {dtype}: allocated {torch.cuda.memory_allocated() / 2**30:.2f} GiB, reserved {torch.cuda.memory_reserved() / 2**30:.2f} GiB", file=sys.stderr)

    # N ranks on N distinct devices, as RCCL sees them: every rank reports its device index and PCI bus id
    rccl = None
    if dist.is_initialized():
        pr = torch.cuda.get_device_properties(local)
        bus = "%04x:%02x:%02x" % (getattr(pr, "pci_domain_id", 0), getattr(pr, "pci_bus_id", -1) & 0xff, getattr(pr, "pci_device_id", 0))
        mine = {"rank": rank, "local_rank": local, "device": local, "pci_bus_id": bus, "name": pr.name, "uuid": str(getattr(pr, "uuid", ""))}
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        rccl = {"backend": dist.get_backend(), "world": dist.get_world_size(), "devices": gathered,
                "distinct_devices": len({(g["pci_bus_id"], g["uuid"]) for g in gathered})}

    if rank == 0:
        clips = B * world * args.steps
        value = clips / dt
        fpc = flops_per_clip(args.workload, d=768 if args.arch == "base" else 384, patch_k=patch[0] * patch[1])
        khz = "32kHz" if args.hires else "16kHz"
        out = {"metric": f"pretrain clips/sec (10s@{khz}, ATST-{args.arch})", "value": round(value, 2), "unit": "clips/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": ("fp32 activations, split-bf16 MFMA GEMMs (parity mode)" if args.precise else "bf16") if args.dtype == "bf16"
                        else DTYPE_NAMES["fp8"], "data": "synthetic",
               "config": {"workload": {"clip6": "ATST-small clip-level, 2 global (10 s) + 4 local (1 s) views",
                                       "clip2": "ATST-small clip-level, 2 views (10 s)",
                                       "frame": "ATST-Frame small, masked frame objective (10 s)"}[args.workload].replace("small", args.arch) +
                          f", synthetic AudioSet-shaped 10s@{khz} waveforms" + (", 128 mel bands, 128 x 8 patches (2001 frames -> 250 tokens)" if args.hires else "") +
                          ", mel front end inside the timed step",
                          "clips_per_gpu": B, "global_batch": B * world, "parallelism": f"dp{world}",
                          "optimizer": "HF-AdamW + EMA teacher (fused)", "drop_path": 0.1},
               "flops_per_clip_G": round(fpc / 1e9, 2), "step_tflops": round(value * fpc / 1e12, 2),
               "mfma_roofline_frac_step": round(value * fpc / 1e12 / (PEAK_BF16_TFLOPS * world), 4),
               "loss": round(loss_val, 5), "roofline": roof, "kernels": kernels[:16]}
        if exclusive:
            out["kernels_exclusive"] = sorted(exclusive.values(), key=lambda r: -r["total_ms"])[:16]   # second stream off (see roofline.concurrency)
        if step_hbm:
            tbs = step_hbm["traffic_bytes_per_step"] / (dt / args.steps) / 1e12
            step_hbm.update({"TB_per_s": round(tbs, 3), "frac_of_8TBs_peak": round(tbs / (PEAK_HBM_GBS / 1e3), 4)})
            out["step_hbm"] = step_hbm
        if also:
            out["also"] = also
        if rccl:
            out["rccl"] = rccl
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
