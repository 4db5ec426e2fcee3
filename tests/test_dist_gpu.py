"""The one-process-per-GPU launch path of bench.py on the single-GPU box: `torch.distributed.run` with one rank, RCCL process
group, and ATST_FORCE_COLLECTIVES=1 so that every exchange of the data-parallel step (SyncBN all-gather / all-reduce, the
flat-gradient all-reduce, the fused monitor all-reduce, barrier, max-over-ranks timing) is actually issued through RCCL.
With one rank the collectives are identities, so the loss must equal the non-distributed run's."""
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
