"""(1) The one-process-per-GPU launch path of bench.py on the single-GPU box: `torch.distributed.run` with one rank, RCCL process
group, and ATST_FORCE_COLLECTIVES=1 so that every exchange of the data-parallel step (SyncBN all-gather / all-reduce, the
flat-gradient all-reduce, the fused monitor all-reduce, barrier, max-over-ranks timing) is actually issued through RCCL.
With one rank the collectives are identities, so the loss must equal the non-distributed run's.
(2) The product path at world size 2: two ranks on shards of a batch == one process on the concatenated batch (loss, flat
gradient, BatchNorm running statistics, parameters after one fused optimizer step), with the bucketed all-reduce overlap
on and off, clip-level and ATST-Frame.  Transport: RCCL when the node has >= 2 GPUs, otherwise gloo with both ranks on
cuda:0 (tests/dist_worker.py) -- so the property is checked on the single-GPU box too.
(3) `python bench.py --gpus 2` launches its own ranks and reports n_gpus 2."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(cmd, extra_env=None):
    env = dict(os.environ)
    env.update(extra_env or {})
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    return json.loads(line)


@pytest.mark.parametrize("workload", ["clip2", "frame"])
def test_torchrun_single_rank_rccl(workload):
    common = ["bench.py", "--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "8", "--workload", workload,
              "--no-cpu-baseline", "--no-profile"]
    plain = run([sys.executable] + common)
    dist = run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                "--master-port", "29541"] + common, {"ATST_FORCE_COLLECTIVES": "1"})
    assert dist["n_gpus"] == 1 and dist["scaling"] == "weak"
    assert abs(dist["loss"] - plain["loss"]) < 2e-3, (dist["loss"], plain["loss"])      # wgrad atomics: not bit-reproducible


def _worker(tmp_path, tag, world, mode, overlap=1, batch=16, ranks=2):
    out = os.path.join(str(tmp_path), f"{tag}.npz")
    base = [os.path.join(ROOT, "tests", "dist_worker.py"), "--mode", mode, "--out", out, "--overlap", str(overlap), "--batch", str(batch),
            "--ranks", str(ranks)]
    if world == 1:
        cmd = [sys.executable] + base
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr",
               "127.0.0.1", "--master-port", str(29560 + overlap + (7 if mode != "clip" else 0) + 20 * (world > 2))] + base
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4")
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    import numpy as np
    return np.load(out)


def _rel(a, b):
    import numpy as np
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


@pytest.mark.parametrize("mode,overlap", [("clip", 1), ("clip", 0), ("frame", 1)])
def test_two_ranks_equal_one_rank_on_the_concatenated_batch(tmp_path, mode, overlap):
    """16 clips per rank.  The two runs differ only in fp32 summation order (Chan-combined vs one-pass BatchNorm statistics,
    wgrad M-splits); each BatchNorm backward amplifies that ~100x through the cancellation dy - mean(dy) - xhat mean(dy xhat)
    (DESIGN.md section 4), more with fewer rows.  Measured (tools/dist_diag.py, profiles/r02_dist_diag.txt): run-to-run
    5e-8; 2 ranks vs 1: whole buffer 1.7e-3, predictor 2.6e-5, projector 8.5e-4 at 16 clips/rank (9.7e-3 / 1.7e-3 /
    1.4e-2 at 4 clips/rank); overlap on == overlap off to the last digit."""
    one = _worker(tmp_path, f"one_{mode}", 1, mode)
    two = _worker(tmp_path, f"two_{mode}_{overlap}", 2, mode, overlap)
    assert int(two["world"]) == 2 and float(two["same"][0]) == 1.0          # ranks hold identical parameters after the step
    assert abs(float(two["loss"][0]) - float(one["loss"][0])) < 2e-4, (two["loss"], one["loss"])
    assert abs(float(two["std_s"]) - float(one["std_s"])) < 1e-4 and abs(float(two["std_t"]) - float(one["std_t"])) < 1e-4
    assert _rel(two["grads"], one["grads"]) < 5e-3, _rel(two["grads"], one["grads"])
    n_pred = 2 * 4096 * 256 + 2 * 4096                                       # predictor = tail of the flat buffer
    assert _rel(two["grads"][-n_pred:], one["grads"][-n_pred:]) < 5e-4
    for k in one.files:
        if k.startswith("bn/") and not k.endswith("num_batches_tracked"):
            assert _rel(two[k], one[k]) < 1e-4, k
    # the first Adam step is lr * g / (|g| + eps): gradient entries near zero turn fp32 noise into +-lr (1e-3 on 2e-2 weights)
    assert _rel(two["params"], one["params"]) < 2e-3 and _rel(two["teacher"], one["teacher"]) < 2e-5


@pytest.mark.parametrize("mode", ["clip", "frame"])
def test_eight_ranks_equal_one_rank_on_the_concatenated_batch(tmp_path, mode):
    """configs[3] readiness on the single-GPU box (BASELINE.json: 8 x MI355X DDP): EIGHT ranks (gloo, all on cuda:0 unless the node has
    8 GPUs), 4 clips each, against one process on the 32-clip batch.  Exercises what world 2 cannot: the bucket cuts with seven
    peers, the SyncBN combine of eight [mean, M2, count] rows, rank -> device mapping, eight DIFFERENT initial seeds aligned by the one
    init broadcast.  Tolerances: the 4-clips-per-rank regime of the two-rank test (BatchNorm-backward cancellation amplifies the fp32
    summation-order difference more with fewer rows per rank)."""
    one = _worker(tmp_path, f"one8_{mode}", 1, mode, batch=4, ranks=8)
    eight = _worker(tmp_path, f"eight_{mode}", 8, mode, 1, batch=4, ranks=8)
    assert int(eight["world"]) == 8 and float(eight["same"][0]) == 1.0      # all ranks hold identical parameters after the step
    g = _rel(eight["grads"], one["grads"])
    n_pred = 2 * 4096 * 256 + 2 * 4096
    gp = _rel(eight["grads"][-n_pred:], one["grads"][-n_pred:])
    print(f"\n[8 ranks vs 1, {mode}] loss {float(eight['loss'][0]):.6f} vs {float(one['loss'][0]):.6f}; flat gradient rel {g:.2e}; predictor {gp:.2e}")
    assert abs(float(eight["loss"][0]) - float(one["loss"][0])) < 3e-4
    assert abs(float(eight["std_s"]) - float(one["std_s"])) < 1e-4 and abs(float(eight["std_t"]) - float(one["std_t"])) < 1e-4
    # measured: clip 2.9e-3 / 3.5e-4, frame 3.4e-4 / 1.3e-5.  The predictor figure is set by a handful of ReLU gates of near-zero pre-activations
    # that the two SyncBN evaluation orders (8 combined partial statistics vs one pass) put on different sides of zero: 2.8e-5 with the head
    # Linears summed in one block per tile, 3.5e-4 with their K split over 24 blocks (csrc/gemm.hip split-K) -- same kernels and the same
    # summation order in both runs either way, only the rounding pattern that meets the gates differs.
    assert g < 5e-3 and gp < 1e-3, (g, gp)
    for k in one.files:
        if k.startswith("bn/") and not k.endswith("num_batches_tracked"):
            assert _rel(eight[k], one[k]) < 1e-4, k
    assert _rel(eight["teacher"], one["teacher"]) < 5e-5


@pytest.mark.parametrize("mode", ["base_fp8", "small_fp8"])
def test_two_ranks_equal_one_rank_base_fp8(tmp_path, mode):
    """BASELINE.json configs[4] is an 8-GPU DDP job: ATST-base with every GEMM of a block on e4m3 operands, two ranks (8 clips each) against one
    process on the 16-clip batch, second step (the first records the delayed scales).  The amax of every quantisation site is MAX-reduced over the
    ranks, so both runs quantise on the same grids (checked on the scales); what is left is summation order meeting the
    e4m3 staircase (a value next to a rounding boundary lands on either side), far inside the fp8 path's own distance from bf16."""
    one = _worker(tmp_path, "one_fp8", 1, mode, batch=8)                    # small_fp8 (round 6): the same at d = 384
    two = _worker(tmp_path, "two_fp8", 2, mode, 1, batch=8)
    assert int(two["world"]) == 2 and float(two["same"][0]) == 1.0          # ranks hold identical parameters after the step
    import numpy as np
    g = _rel(two["grads"], one["grads"])
    sg = float(np.max(np.abs(two["g8_scale"] * 2.0 / one["g8_scale"] - 1.0))); sa = float(np.max(np.abs(two["f8a_scale"] / one["f8a_scale"] - 1.0)))
    print(f"\n[2 ranks vs 1, {mode}] loss {float(two['loss'][0]):.6f} vs {float(one['loss'][0]):.6f}; flat gradient rel {g:.2e}; scales: gradient sites {sg:.1e}, forward sites {sa:.1e}")
    print("  gradient-site scale ratios (2 ranks x 2 / 1 rank):", np.array2string(two["g8_scale"].reshape(-1) * 2.0 / one["g8_scale"].reshape(-1), precision=4))
    # same quantisation grids: a rank's activation gradients are world x the single process's (its loss is the mean over ITS clips; DDP averages the
    # parameter gradients afterwards), so its gradient scales are 1 / world of them -- scale x value, i.e. the e4m3 code, is the same; forward scales equal
    # (a site's amax is ONE element: the summation-order difference between the runs moves it by what the e4m3 staircase of its inputs allows -- measured
    # per site: base <= 8.9e-3 (a du site) ; small 6.8e-3 with the LayerNorms as separate passes and 2.0e-2 with them inside the GEMM epilogues, both at
    # block 0's dqkv site, which the attention backward posts; a missing MAX-reduce shows as ranks holding different parameters -- `same` above)
    assert sg < 4e-2 and sa < 1e-2
    assert abs(float(two["loss"][0]) - float(one["loss"][0])) < 2e-3
    assert g < 4e-2, g                                                     # measured 2.1e-2 (loss 3e-6 apart, forward scales identical, gradient scales 1.5e-3 apart)
    assert _rel(two["teacher"], one["teacher"]) < 1e-4


def test_two_ranks_ragged_frame_rows(tmp_path):
    """Per-rank masked-row counts differ (SyncBN count-weighted combine with a device-side total, ragged gather): DDP then
    averages per-rank mean-loss gradients, which is not the concatenated-batch gradient -- checked here: the step runs,
    every rank ends with identical parameters, and the loss is close to the concatenated run's."""
    one = _worker(tmp_path, "one_fr", 1, "frame_ragged", batch=4)
    two = _worker(tmp_path, "two_fr", 2, "frame_ragged", 1, batch=4)
    assert float(two["same"][0]) == 1.0
    assert abs(float(two["loss"][0]) - float(one["loss"][0])) < 2e-2
    for k in one.files:
        if k.startswith("bn/") and not k.endswith("num_batches_tracked"):
            assert _rel(two[k], one[k]) < 1e-4, k                           # SyncBN statistics ARE the concatenated batch's


def test_bench_launches_its_own_ranks():
    import torch
    extra = [] if torch.cuda.device_count() >= 2 else ["--backend", "gloo"]
    out = run([sys.executable, "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "8", "--workload", "clip2",
               "--no-cpu-baseline", "--no-profile"] + extra)
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 16 and out["config"]["parallelism"] == "dp2"
