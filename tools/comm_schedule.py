#!/usr/bin/env python3
"""The collectives of ONE training step, in issue order: which tensor, how many bytes, on which HIP stream, and how much backward
work is still to be enqueued behind each gradient bucket.  Runs one rank through RCCL (ATST_FORCE_COLLECTIVES=1) -- a one-GPU box
cannot time xGMI, but the schedule (count, sizes, placement) is the same at any world size; the 8-GPU curve is SCALE_rNN.json.
usage (GPU box): python tools/comm_schedule.py [clip6|clip2|frame] [out.txt]"""
import os, sys
os.environ["ATST_FORCE_COLLECTIVES"] = "1"
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
import torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
workload = sys.argv[1] if len(sys.argv) > 1 else "clip6"
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from audiossl_amd.engine import AtstEngine
from audiossl_amd import hip
log, phase = [], ["init"]
real = dist.all_reduce
main = torch.cuda.current_stream()
launched = [0]
lib = hip.load()
def counting_call(name, *a):                    # count kernel-launching library calls to place the collectives on the step's timeline
    launched[0] += 1
    return orig_call(name, *a)
orig_call = hip.call; hip.call = counting_call
def logged(t, *a, **k):
    st = torch.cuda.current_stream()
    log.append((phase[0], t.numel() * t.element_size(), "comm stream" if st != main else "main stream", launched[0]))
    return real(t, *a, **k)
dist.all_reduce = logged
from audiossl_amd import engine as E
_br, _b = E.EncoderPass.backward_range, E.EncoderPass.backward
def br(self, lo, hi):
    log.append((f"  backward of blocks [{lo}, {hi}) of the {self.S}-sequence pass" + (" + final LayerNorm" if hi == self.eng.depth else "") + (" + token stage" if lo == 0 else ""), 0, "main stream", launched[0]))
    return _br(self, lo, hi)
def bb(self):
    log.append((f"  backward of the whole {self.S}-sequence pass", 0, "main stream", launched[0]))
    return _b(self)
E.EncoderPass.backward_range, E.EncoderPass.backward = br, bb
frame = workload == "frame"; B = 32
ncrops = 6 if workload == "clip6" else 2
eng = AtstEngine("small", frame=frame, ncrops=ncrops)
eng.init_weights(seed=0); eng.broadcast_parameters()
mels = [torch.randn(B, 1, 64, 1001, device="cuda").clamp(-1, 1) for _ in range(2)]
lens = [torch.full((B,), 1001)] * 2
masks = None
if ncrops == 6:
    mels += [torch.randn(B, 1, 64, 101, device="cuda").clamp(-1, 1) for _ in range(4)]; lens += [torch.full((B,), 101)] * 4
if frame:
    import numpy as np
    from audiossl_amd.methods.atstframe.random_mask import block_mask
    rs = np.random.RandomState(0); m = torch.from_numpy(np.stack([block_mask(250, 0.65, 5, rng=rs) for _ in range(B)])); masks = [m, m]
for it in range(2):
    log.clear(); launched[0] = 0
    phase[0] = "forward"; eng.forward(mels, lens, masks)
    n_fwd = launched[0]
    phase[0] = "backward"; eng.backward()
    n_bwd = launched[0]
    phase[0] = "optimizer"; eng.allreduce_grads(); eng.optimizer_step(1e-3, 0.04, 0.99)
torch.cuda.synchronize()
total = eng.g32.numel() * 4
out = [f"# collectives of one {workload} step (ATST-small, {eng.layout.n_student} student parameters, flat fp32 gradient {total / 1e6:.1f} MB), in issue order;",
       f"# encoder backward calls interleaved to show what each gradient bucket hides under.  bucket cuts (blocks, descending): {eng.bucket_cuts()}"]
grad_seen = 0
for ph, nbytes, st, pos in log:
    if nbytes == 0:
        out.append(ph); continue
    if nbytes >= 1e6:
        grad_seen += nbytes
        what = f"all-reduce of a gradient bucket, {nbytes / 1e6:5.1f} MB = {nbytes / total * 100:4.1f} % of the gradient"
    elif ph == "forward":
        what = "SyncBN statistics (teacher + student projector in one call)" if nbytes > 40000 else ("SyncBN statistics (predictor)" if nbytes > 9000 else "monitor sums")
    else:
        what = "SyncBN backward sums"
    out.append(f"{ph:9s} {st:12s} {nbytes:10d} B   {what}")
out.append(f"# gradient bytes reduced: {grad_seen / 1e6:.1f} MB of {total / 1e6:.1f} MB; only the LAST bucket has no backward work behind it (exposed tail).")
txt = "\n".join(out); print(txt)
if len(sys.argv) > 2: open(sys.argv[2], "w").write(txt + "\n")
dist.destroy_process_group()
