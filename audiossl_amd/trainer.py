"""Minimal trainer with Lightning 2.2's automatic-optimisation hook order, used when
pytorch_lightning is not installed:  schedule()+training_step -> zero_grad -> backward -> on_after_backward (grad
all-reduce) -> optimizer.step -> global_step += 1 -> on_train_batch_end (EMA, index = post-increment step).
Writes / reads Lightning-format checkpoint dicts (keys consumed by the reference's downstream tools:
``state_dict`` with ``model.`` prefix, ``hyper_parameters``, ``pytorch-lightning_version``, ``global_step``, ``epoch``,
``optimizer_states``; ref: audiossl/methods/atst/train.py:25-35, downstream/train_freeze.py:27-35)."""
from __future__ import annotations

import os
import time

import torch
import torch.distributed as dist

LIGHTNING_VERSION = "2.2.1"                           # setup.cfg:22 of the reference


def save_checkpoint(path, module, optimizer=None, epoch=0):
    """epoch: zero-based index of the last FINISHED epoch (Lightning's convention); -1 = none finished yet (a max_steps stop inside
    epoch 0): stored as `epoch` 0 for Lightning-format consumers, with the exact count in `epochs_finished` for our own resume."""
    sd = {k: v.detach().cpu() for k, v in module.state_dict().items()}
    eng = module.model.engine
    ckpt = {"state_dict": sd, "hyper_parameters": dict(getattr(module, "hparams", {})), "global_step": int(module.global_step),
            "epoch": max(int(epoch), 0), "epochs_finished": int(epoch) + 1,
            "pytorch-lightning_version": LIGHTNING_VERSION, "loops": {}, "callbacks": {},
            "lr_schedulers": [],
            "optimizer_states": [{"flat_exp_avg": eng.m32.cpu(), "flat_exp_avg_sq": eng.v32.cpu(), "step": eng.opt_step,
                                  "layout": "audiossl_amd.FlatLayout", "fp8_dgrad": eng.fp8_state()}]}
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    torch.save(ckpt, path)


def load_checkpoint(path, module, strict=True):
    ckpt = torch.load(path, map_location="cpu", weights_only=False)
    assert "pytorch-lightning_version" in ckpt, "not a Lightning-format checkpoint"
    module.load_state_dict(ckpt["state_dict"], strict=strict)
    module.model.engine.sync_shadows(force=True)
    module.global_step = int(ckpt.get("global_step", 0))
    st = (ckpt.get("optimizer_states") or [None])[0]
    if st and st.get("layout") == "audiossl_amd.FlatLayout":
        eng = module.model.engine
        eng.m32.copy_(st["flat_exp_avg"]); eng.v32.copy_(st["flat_exp_avg_sq"]); eng.opt_step = int(st["step"])
        eng.load_fp8_state(st.get("fp8_dgrad"))                    # delayed-scaling scales + amax window: a resumed run quantises as the saved one did
    return ckpt


class Trainer:
    def __init__(self, max_steps=-1, max_epochs=None, default_root_dir=None, every_n_epochs=20, log_every_n_steps=50,
                 batch_hook=None, **_ignored):
        self.max_steps, self.max_epochs = max_steps, max_epochs
        self.root, self.every_n_epochs, self.log_every = default_root_dir, every_n_epochs, log_every_n_steps
        self.batch_hook = batch_hook                       # e.g. ATSTBatchViews: waveform views -> mel views on the GPU
        self.optimizers = []
        self.rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        self.history = []

    def fit(self, model, datamodule=None, train_dataloaders=None, ckpt_path=None):
        model.trainer = self
        self.optimizers = model.configure_optimizers()
        opt = self.optimizers[0]
        epoch = 0
        # Resume: RANK 0 decides (its view of the file system is the only one that counts), every rank learns the decision and
        # the counters through one broadcast, and all replicas are then aligned by ONE broadcast of rank 0's tensors -- never a
        # per-rank os.path.exists() in front of a collective (ranks that do not see the file would deadlock the others).
        eng = model.model.engine
        state = [False, 0, int(model.global_step), None]
        if self.rank == 0 and ckpt_path and os.path.exists(ckpt_path):
            try:
                ckpt = load_checkpoint(ckpt_path, model)
                # Lightning stores the zero-based index of the last FINISHED epoch; `epochs_finished` (ours) is exact when no epoch was finished
                done_epochs = int(ckpt["epochs_finished"]) if "epochs_finished" in ckpt else int(ckpt.get("epoch", -1)) + 1
                state = [True, done_epochs, int(model.global_step), None]
            except Exception as e:                                # a corrupt / partial file on rank 0: every rank must learn it, not hang in the broadcast
                state = [False, 0, int(model.global_step), f"{type(e).__name__}: {e}"]
        if self.world > 1:
            dist.broadcast_object_list(state, src=0)
        if state[3] is not None:
            raise RuntimeError(f"resume from {ckpt_path} failed on rank 0: {state[3]}")
        resumed, epoch, model.global_step = bool(state[0]), int(state[1]), int(state[2])
        eng.broadcast_parameters(optimizer_state=resumed)
        loader = train_dataloaders if train_dataloaders is not None else datamodule.train_dataloader(self.rank, self.world)
        dev = model.model.engine.device
        t0, step0 = time.time(), model.global_step
        while True:
            if hasattr(getattr(loader, "sampler", None), "set_epoch"):
                loader.sampler.set_epoch(epoch)
            exhausted = False
            for batch_idx, batch in enumerate(loader):
                inputs, labels = batch[0], batch[1]
                # clip ATST: (views, lengths) (methods/atst/model.py:26) ; ATST-Frame: (views, lengths, masks) (atstframe/model.py:120)
                views, lengths, masks = (inputs[0], inputs[1], inputs[2] if len(inputs) > 2 else None)
                views = [v.to(dev, non_blocking=True) for v in views]
                if self.batch_hook is not None and views[0].dim() == 3:            # [B,1,n] waveforms -> mel views
                    views = self.batch_hook(views, lengths)
                step_in = (views, lengths) if masks is None else (views, lengths, masks)
                loss = model.training_step((step_in, labels), batch_idx)
                opt.zero_grad()
                loss.backward()
                model.on_after_backward()
                opt.step()
                model.global_step += 1
                model.on_train_batch_end(loss, batch, batch_idx)
                if self.rank == 0 and model.global_step % self.log_every == 0:
                    rec = {k: (float(v.detach()) if torch.is_tensor(v) else v) for k, v in model.logged.items()}
                    rec["it/s"] = (model.global_step - step0) / (time.time() - t0)
                    self.history.append(rec)
                    print(" ".join(f"{k}={v:.5g}" if isinstance(v, float) else f"{k}={v}" for k, v in rec.items()), flush=True)
                if 0 < self.max_steps <= model.global_step:
                    exhausted = hasattr(loader, "__len__") and batch_idx + 1 >= len(loader)     # stopped on the epoch's last batch
                    break
            else:
                exhausted = True
            # `epoch` counts FINISHED epochs: a max_steps stop in the middle of an epoch does not finish it, so a resume repeats
            # that epoch from its first batch (sampler epoch unchanged) instead of skipping its remainder
            epoch += 1 if exhausted else 0
            done = (0 < self.max_steps <= model.global_step) or (self.max_epochs and epoch >= self.max_epochs)
            if self.root and self.rank == 0:
                if exhausted and epoch % self.every_n_epochs == 0:
                    save_checkpoint(os.path.join(self.root, f"checkpoint-epoch={epoch - 1:05d}.ckpt"), model, opt, epoch - 1)
                if done or (exhausted and epoch % self.every_n_epochs == 0):
                    save_checkpoint(os.path.join(self.root, "last.ckpt"), model, opt, epoch - 1)
            if done:
                break
        return model
