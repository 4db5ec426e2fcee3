"""World-size-2 and -8 (gloo, CPU) tests of the cross-rank logic in audiossl_amd/parallel.py: a sharded evaluation must equal the
1-rank evaluation on the concatenated batch -- which is what DDP-mean + SyncBatchNorm guarantee; world 8 is the
geometry of BASELINE.json configs[3] (one node, 8 GPUs)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from audiossl_amd import parallel as P
    torch.manual_seed(0)
    N = 64
    cuts = [0, 9, 22] if world == 2 else [0, 3, 9, 10, 17, 22, 30, 33, 41]     # ragged shards (ATST-Frame: masked-row counts differ per rank)
    R = cuts[-1]
    h = torch.randn(R, N, dtype=torch.float64) * 2 + 0.5
    gamma, beta = torch.rand(N, dtype=torch.float64) + 0.5, torch.randn(N, dtype=torch.float64) * 0.1
    w_out = torch.randn(N, dtype=torch.float64)
    lo, hi = cuts[rank], cuts[rank + 1]
    hl = h[lo:hi]
    # ---- forward statistics
    mean_l = hl.mean(0); m2_l = ((hl - mean_l) ** 2).sum(0)
    mean, m2, count = P.combine_bn_stats(mean_l, m2_l, hl.shape[0])
    ok = torch.allclose(mean, h.mean(0)) and torch.allclose(m2 / count, h.var(0, unbiased=False)) and float(count) == R
    # ---- backward: loss = mean over ALL rows of relu(bn(h)) . w_out ; each rank holds d(loss)/dy for its rows
    rstd = torch.rsqrt(m2 / count + 1e-5)
    hr = h.clone().requires_grad_(True); gr = gamma.clone().requires_grad_(True); br = beta.clone().requires_grad_(True)
    y = torch.relu(torch.nn.functional.batch_norm(hr, None, None, gr, br, True, 0.1, 1e-5))
    (y @ w_out).mean().backward()
    xh = (hl - mean) * rstd
    dy = (w_out / R).expand_as(hl) * ((xh * gamma + beta) > 0)
    s1_l, s2_l = dy.sum(0), (dy * xh).sum(0)
    s1, s2 = P.allreduce_bn_backward_sums(s1_l, s2_l)
    dh = gamma * rstd * (dy - s1 / count - xh * s2 / count)
    ok = ok and torch.allclose(dh, hr.grad[lo:hi]) and torch.allclose(s1, br.grad) and torch.allclose(s2, gr.grad)
    # ---- DDP gradient mean: local-mean-loss gradients summed then scaled by 1/world == global-mean-loss gradient
    #      (equal shard sizes, as drop_last + DistributedSampler give)
    X = torch.randn(16, 5, dtype=torch.float64); t = torch.randn(16, dtype=torch.float64); w = torch.randn(5, dtype=torch.float64)
    idx = list(P.shard(16, rank, world))
    g_local = (2 * (X[idx] @ w - t[idx])[:, None] * X[idx]).mean(0)
    flat = g_local.clone()
    summed = P.allreduce_sum_(flat)
    g_full = (2 * (X @ w - t)[:, None] * X).mean(0)
    ok = ok and summed and torch.allclose(flat / world, g_full)
    # ---- two independent BatchNorms in ONE exchange (teacher + student projector of a step): same numbers as one by one
    h2 = torch.randn(R, N, dtype=torch.float64) * 0.7 - 1.0
    h2l = h2[lo:hi]
    m_b = h2l.mean(0); q_b = ((h2l - m_b) ** 2).sum(0)
    (ma, qa, ca), (mb, qb, cb) = P.combine_bn_stats_multi([(mean_l, m2_l, hl.shape[0]), (m_b, q_b, h2l.shape[0])])
    ok = ok and torch.allclose(ma, mean) and torch.allclose(qa, m2) and float(ca) == R
    ok = ok and torch.allclose(mb, h2.mean(0)) and torch.allclose(qb / cb, h2.var(0, unbiased=False)) and float(cb) == R
    # ---- fused monitor all-reduce
    stats = torch.arange(8, dtype=torch.float64).view(4, 2) * (rank + 1)
    st, ns, nt = P.allreduce_monitor_sums(stats, 4.0 + rank, 2.0)
    tri = world * (world + 1) // 2
    ok = ok and torch.equal(st, torch.arange(8, dtype=torch.float64).view(4, 2) * tri) and float(ns) == 4.0 * world + tri - world and float(nt) == 2.0 * world
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_sharded_ranks_equal_one_rank(world):
    ctx = mp.get_context("spawn")
    q, port = ctx.Queue(), _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(r, True) for r in range(world)]


def test_single_process_helpers_are_identity():
    from audiossl_amd import parallel as P
    m, v = torch.randn(8), torch.rand(8)
    assert P.combine_bn_stats(m, v, 5)[2] == 5.0 and P.combine_bn_stats(m, v, 5)[0] is m
    assert P.allreduce_sum_(torch.zeros(3)) is False
    (a, b, c), = P.combine_bn_stats_multi([(m, v, 5)])
    assert a is m and b is v and c == 5.0
    assert list(P.shard(10, 1, 4)) == [1, 5] and P.world_size() == 1
    s = torch.tensor([2.0, 4.0]); q = torch.tensor([3.0, 9.0])
    assert torch.isfinite(P.feature_std(s, q, 4.0))
