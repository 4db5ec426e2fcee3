"""Cross-rank pieces of the data-parallel step (one process per GPU, torch.distributed: RCCL over xGMI on the GPU box,
gloo in the CPU tests).  The path shards by clips; the only exchanges are

  * SyncBatchNorm statistics of the three head BatchNorms (Trainer(sync_batchnorm=True), methods/atst/train.py:22):
    forward  all_gather [mean, M2, count]       -> count-weighted parallel-variance combine (ragged counts allowed)
    backward all_reduce [sum_dy, sum_dy_xhat]   -> global dx ; parameter gradients stay local (DDP averages them)
  * one all-reduce (sum) of the flat student-gradient buffer; the 1/world mean is folded into the optimizer kernel
  * one fused all-reduce of the monitor sums (the reference issues six: models/atst/byol.py:42-53)

These functions are device-agnostic torch code on tiny tensors (plumbing); everything heavy stays in the HIP library."""
from __future__ import annotations

from typing import Tuple

import os

import torch
import torch.distributed as dist


def world_size() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def _collective() -> bool:
    """True when the exchanges must be issued.  ATST_FORCE_COLLECTIVES=1 issues them at world size 1 as well, so that the
    RCCL code path can be exercised on a single-GPU box (tests/test_dist_gpu.py)."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or os.environ.get("ATST_FORCE_COLLECTIVES") == "1"


def _gather_rows(pack: torch.Tensor) -> torch.Tensor:
    """Every rank's `pack` (1-D, equal length) as the rows of a [world, n] tensor.  RCCL: one all_gather_into_tensor (n floats sent per
    rank).  Other transports (gloo has no all_gather for device tensors): the same table as an all-reduce of zeros with this
    rank's row filled -- world x the bytes plus a fill, which is what every step paid before round 4."""
    out = pack.new_empty(world_size(), pack.numel())
    if dist.get_backend() == "nccl":
        dist.all_gather_into_tensor(out, pack.contiguous())
    else:
        out.zero_()
        out[dist.get_rank()] = pack
        dist.all_reduce(out)
    return out


def combine_bn_stats(mean: torch.Tensor, m2: torch.Tensor, count: float):
    """Local per-feature (mean, M2 = sum (x-mean)^2, row count) -> global (mean, M2, count) over all ranks
    (Chan et al. parallel variance).  Identity at world size 1 (count stays a python float); across ranks the global
    count is returned as a 0-dim tensor ON THE DEVICE OF `mean` -- per-rank row counts are ragged for ATST-Frame, and
    reading the total back would drain the launch queue once per BatchNorm per step."""
    if not _collective():
        return mean, m2, float(count)
    n = mean.numel()
    pack = torch.cat([mean, m2, mean.new_full((1,), float(count))])                 # fill kernel: no host->device copy, no sync
    allp = _gather_rows(pack)                                                       # [world, 2n+1]
    means, m2s, cnts = allp[:, :n], allp[:, n:2 * n], allp[:, -1:]
    total = cnts.sum()
    gmean = (means * cnts).sum(0) / total
    gm2 = (m2s + cnts * (means - gmean) ** 2).sum(0)
    return gmean, gm2, total


def combine_bn_stats_multi(items):
    """combine_bn_stats for several independent BatchNorms in ONE collective.  items: [(mean, M2, count), ...] -> same structure
    with global values.  (Small collectives are latency-bound: the teacher and student projector exchanges of a step ride together.)"""
    if not _collective():
        return [(m, q, float(c)) for m, q, c in items]
    packs = [torch.cat([m, q, m.new_full((1,), float(c))]) for m, q, c in items]
    sizes = [p_.numel() for p_ in packs]
    pack = torch.cat(packs)
    allp = _gather_rows(pack)
    out, off = [], 0
    for (m, _, _), sz in zip(items, sizes):
        n = m.numel()
        blk = allp[:, off:off + sz]
        means, m2s, cnts = blk[:, :n], blk[:, n:2 * n], blk[:, -1:]
        total = cnts.sum()
        gmean = (means * cnts).sum(0) / total
        gm2 = (m2s + cnts * (means - gmean) ** 2).sum(0)
        out.append((gmean, gm2, total))
        off += sz
    return out


def allreduce_bn_backward_sums(sum_dy: torch.Tensor, sum_dy_xhat: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """Global sums for the BatchNorm input gradient; the local sums remain the (to-be-averaged) gamma/beta gradients."""
    if not _collective():
        return sum_dy, sum_dy_xhat
    n = sum_dy.numel()
    pack = torch.cat([sum_dy, sum_dy_xhat])
    dist.all_reduce(pack)
    return pack[:n].contiguous(), pack[n:].contiguous()


def allreduce_monitor_sums(stats: torch.Tensor, n_student: float, n_teacher: float):
    """[4, D] column sums / square sums of normalised student & teacher rows + the two row counts, one all-reduce.
    Across ranks the counts come back as 0-dim tensors on the device of `stats` (no read-back)."""
    if not _collective():
        return stats, float(n_student), float(n_teacher)
    pack = torch.cat([stats.reshape(-1), stats.new_full((1,), float(n_student)), stats.new_full((1,), float(n_teacher))])
    dist.all_reduce(pack)
    return pack[:-2].view_as(stats), pack[-2], pack[-1]


def feature_std(sums: torch.Tensor, sq_sums: torch.Tensor, n: float) -> torch.Tensor:
    """mean_d sqrt(unbiased var_d + 1e-6) from global sums.  ref: audiossl/models/atst/byol.py:42-53."""
    return torch.sqrt(sq_sums / (n - 1) - sums ** 2 / (n * (n - 1)) + 1e-6).mean()


def allreduce_sum_(flat_grad: torch.Tensor) -> bool:
    """In-place sum of the flat gradient buffer over ranks; returns True when a reduction happened."""
    if not _collective():
        return False
    dist.all_reduce(flat_grad)
    return True


def shard(n_items: int, rank: int, world: int) -> range:
    """Strided shard of a non-shuffled index list, like the DistributedSampler Lightning injects."""
    return range(rank, n_items - n_items % world, world)
