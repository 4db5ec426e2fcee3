"""Worker of tests/test_dist_gpu.py (NOT a test module).  Runs ONE training step of the product path either
  * as a single process on the concatenated batch (--world 1), or
  * as `--world W` ranks (started by torch.distributed.run), each on its shard of the same batch,
and writes loss / flat gradient / BatchNorm running statistics / parameters after one optimizer step to --out (rank 0).
Transport: RCCL ("nccl") when the node has >= W GPUs, otherwise gloo with all ranks sharing cuda:0 -- the collectives,
SyncBN, bucketed overlap, and optimizer code paths are the same, so the sharded == concatenated property can be checked
on a single-GPU box.  Ranks deliberately initialise with DIFFERENT seeds: the init broadcast must make them equal."""
import argparse
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", default="clip", choices=["clip", "frame", "frame_ragged", "base_fp8", "small_fp8"])
    ap.add_argument("--out", required=True)
    ap.add_argument("--overlap", type=int, default=1)
    ap.add_argument("--batch", type=int, default=4, help="clips per rank")
    ap.add_argument("--ranks", type=int, default=2, help="number of shards the batch is cut into")
    args = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    ndev = torch.cuda.device_count()
    torch.cuda.set_device(rank % ndev)
    if world > 1:
        backend = "nccl" if ndev >= world else "gloo"
        dist.init_process_group(backend, rank=rank, world_size=world)
    from audiossl_amd.engine import AtstEngine
    from oracle import atst_oracle as O                       # seeded input recipes only (test infrastructure)

    frame = args.mode in ("frame", "frame_ragged")
    fp8 = args.mode in ("base_fp8", "small_fp8")                             # BASELINE.json configs[4] under DDP: ATST-base, every GEMM of a block on e4m3 operands
    depth, Bt = (2 if fp8 else 4), args.batch * args.ranks    # total clips
    if fp8:
        eng = AtstEngine("base" if args.mode == "base_fp8" else "small", depth=depth, ncrops=2, drop_path_rate=0.1, fp8=True)   # small_fp8: the all-e4m3 step at d = 384 (round 6)
    else:
        eng = AtstEngine("small", frame=frame, depth=depth, ncrops=2 if frame else 3, drop_path_rate=0.1)
    eng.overlap_comm = bool(args.overlap)
    eng.grad_buckets = 2
    eng.init_weights(seed=100 + rank)                          # rank-dependent on purpose
    eng.broadcast_parameters()                                 # the ONE explicit DDP-init collective: every rank takes rank 0's replica
    if world == 1:                                             # the single-process run uses what rank 0 broadcasts
        eng.init_weights(seed=100)
    # ---- the full batch, identical in every process (CPU, seeded)
    if frame:
        widths = [1001, 1001]
        mels = [O.recipe_mel(Bt, 1001, seed=11), O.recipe_mel(Bt, 1001, seed=12)]
        lens = [torch.tensor([1001, 900, 640, 1001, 801, 1001, 500, 1001][:Bt]).repeat((Bt + 7) // 8)[:Bt]] * 2
        rs = np.random.RandomState(5)
        from audiossl_amd.methods.atstframe.random_mask import block_mask
        m = np.stack([block_mask(250, 0.65, 5, rng=rs) for _ in range(Bt)])
        if args.mode == "frame":                               # equal masked-row totals per shard: rank r re-uses shard 0's masks
            per = Bt // args.ranks
            m = np.concatenate([m[:per]] * args.ranks)
            lens = [torch.cat([lens[0][:per]] * args.ranks)] * 2
        masks = [torch.from_numpy(m)] * 2
    else:
        widths = [1001, 1001] if fp8 else [1001, 1001, 101]
        mels = [O.recipe_mel(Bt, w, seed=11 + i) for i, w in enumerate(widths)]
        lens = [torch.full((Bt,), 1001), torch.tensor([1001, 900, 640, 1001, 801, 1001, 500, 1001] * Bt)[:Bt], torch.full((Bt,), 101)][:len(widths)]
        masks = None
    g = torch.Generator().manual_seed(3)
    # DropPath keep masks for the whole batch, per width group: [depth, 2, n_views_in_group * Bt]
    def keep_full(nv):
        rates = torch.linspace(0, 0.1, depth).view(-1, 1, 1)
        return torch.floor((1 - rates) + torch.rand(depth, 2, nv * Bt, generator=g))
    if frame or fp8:
        keep_t, keep_s = [keep_full(2)], [keep_full(2)]
    else:
        keep_t, keep_s = [keep_full(2)], [keep_full(2), keep_full(1)]
    # ---- shard: clips [lo, hi) of every view
    lo, hi = (0, Bt) if world == 1 else (rank * args.batch, (rank + 1) * args.batch)
    def shard_keep(k, nv):
        idx = torch.cat([torch.arange(lo, hi) + v * Bt for v in range(nv)])
        return k[:, :, idx].contiguous()
    mels_r = [x[lo:hi].contiguous().cuda() for x in mels]
    lens_r = [l[lo:hi] for l in lens]
    masks_r = None if masks is None else [mk[lo:hi] for mk in masks]
    kt = [shard_keep(keep_t[0], 2)]
    ks = [shard_keep(keep_s[0], 2)] + ([shard_keep(keep_s[1], 1)] if not (frame or fp8) else [])

    if fp8:                                                    # step 1 records the delayed scales (amax MAX-reduced over the ranks), step 2 -- reported -- runs in e4m3
        # (no optimizer step in between: the two steps then see the same weights, and sharded vs concatenated differ by summation order only)
        eng.forward(mels_r, lens_r, masks_r, keep_teacher=kt, keep_student=ks)
        eng.backward()
        eng.allreduce_grads()
        eng.g32.zero_(); eng._grads_summed = False
        assert eng.fp8_bwd_state == 2 and eng.fp8_wgrad_mode() == 2
    loss, std_s, std_t = eng.forward(mels_r, lens_r, masks_r, keep_teacher=kt, keep_student=ks)
    eng.backward()
    eng.allreduce_grads()
    grads = eng.g32.clone()
    if eng._grads_summed:
        grads /= world
    eng.optimizer_step(1e-3, 0.04, 0.99)
    torch.cuda.synchronize()
    lossv = loss.detach().clone().reshape(1)
    if world > 1:
        dist.all_reduce(lossv)
        lossv /= world
        # every rank must hold identical parameters after the step
        ref = eng.p32.clone()
        dist.broadcast(ref, 0)
        same = torch.tensor([float(torch.equal(ref, eng.p32))], device=ref.device)
        dist.all_reduce(same, op=dist.ReduceOp.MIN)
    else:
        same = torch.ones(1)
    if rank == 0:
        bn = {f"bn/{k}/{b}": t.detach().float().cpu().numpy() for k, d in eng.bn_buffers.items() for b, t in d.items()}
        np.savez(args.out, loss=lossv.cpu().numpy(), std_s=float(std_s), std_t=float(std_t), grads=grads.cpu().numpy(),
                 params=eng.p32.cpu().numpy(), teacher=eng.t32.cpu().numpy(), same=same.cpu().numpy(), world=world,
                 backend=(dist.get_backend() if world > 1 else "none"),
                 g8_scale=(eng.g8_scale.cpu().numpy() if fp8 else np.zeros(1)), f8a_scale=(eng.f8a_scale.cpu().numpy() if fp8 else np.zeros(1)), **bn)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
