"""``ATSTLightningModule`` with the reference's constructor, hooks, schedule tables and argparse surface
(audiossl/methods/atst/model.py:6-64), running on the HIP engine.  pytorch_lightning is optional: if it is importable
the class derives from LightningModule, otherwise from a small stand-in that audiossl_amd.trainer.Trainer drives with
the same hook order (Lightning 2.2 automatic optimisation)."""
from __future__ import annotations

import torch

from ...models.atst import ATST
from ...optim import FusedHFAdamW
from ...utils.common import cosine_scheduler_step, get_params_groups

try:                                                    # pragma: no cover - not installed in the build image
    from pytorch_lightning import LightningModule as _Base
    HAVE_LIGHTNING = True
except Exception:                                       # noqa: BLE001
    HAVE_LIGHTNING = False

    class _Base(torch.nn.Module):
        """Minimal LightningModule stand-in: global_step / trainer / log / save_hyperparameters."""

        def __init__(self):
            super().__init__()
            self.trainer = None
            self.global_step = 0
            self.logged = {}
            self.hparams = {}

        def log(self, name, value, **kw):
            self.logged[name] = value

        def on_after_backward(self):
            pass

        @classmethod
        def load_from_checkpoint(cls, checkpoint_path, map_location=None, strict=True, **kwargs):
            """Lightning-format checkpoint -> module (keys as written by Lightning's ModelCheckpoint and by
            audiossl_amd.trainer.save_checkpoint; ref consumer: methods/atst/downstream/train_freeze.py:27-35)."""
            from ...trainer import load_checkpoint
            ckpt = torch.load(checkpoint_path, map_location="cpu", weights_only=False)
            hp = dict(ckpt.get("hyper_parameters", {}))
            hp.update(kwargs)
            module = cls(**hp)
            load_checkpoint(checkpoint_path, module, strict=strict)
            return module

        def save_hyperparameters(self, **kw):
            import inspect
            frame = inspect.currentframe().f_back
            args = inspect.getargvalues(frame)
            hp = {k: args.locals[k] for k in args.args if k != "self"}
            if args.keywords and args.keywords in args.locals:
                hp.update(args.locals[args.keywords])
            self.hparams = hp


class ATSTLightningModule(_Base):
    """ref: audiossl/methods/atst/model.py:6-23."""
    _model_cls = ATST
    _frame = False

    def __init__(self, arch="small", learning_rate: float = 5e-4, warmup_steps=1300, max_steps=39000, ema=0.99, **kwargs):
        super().__init__()
        self.model = self._build_model(arch, kwargs)
        self.learning_rate = learning_rate
        self.warmup_steps = warmup_steps
        self.max_steps = max_steps
        self.ema_scheduler = cosine_scheduler_step(ema, 1, max_steps, 0)
        self.wd_scheduler = cosine_scheduler_step(0.04, 0.4, max_steps, 0)
        self.mylr_scheduler = cosine_scheduler_step(learning_rate, 1e-6, max_steps, warmup_steps)
        self.save_hyperparameters()

    def _build_model(self, arch, kwargs):
        return ATST(arch=arch)                          # the reference passes no kwargs (model.py:16)

    # ---- hooks, in Lightning's order --------------------------------------------------------------------------------
    def training_step(self, batch, batch_idx):
        """ref: model.py:24-34."""
        self.schedule()
        (melspecs, lengths), _ = batch
        if melspecs[0].dim() == 3:                      # [B,1,n] waveform crops from the worker-side transform: mel / Mixup / RRC on the GPU
            if getattr(self, "_batch_views", None) is None:
                from .transform import ATSTBatchViews
                self._batch_views = ATSTBatchViews(device=self.model.engine.device)
            melspecs = self._batch_views([m.to(self.model.engine.device, non_blocking=True) for m in melspecs], lengths)
        loss, std_cls_s, std_cls_t = self.model(melspecs, lengths)
        self.log("loss", loss, prog_bar=True, logger=True)
        self.log("std_cls_t", std_cls_t, prog_bar=True, logger=True)
        self.log("std_cls_s", std_cls_s, prog_bar=True, logger=True)
        self.log("ema", self.ema_scheduler[self._idx(self.global_step)], prog_bar=True, logger=True)
        self.log("step", self.global_step, prog_bar=True, logger=True)
        return loss

    def _idx(self, k):
        return min(int(k), self.max_steps - 1)          # the reference indexes one past the table on the last EMA

    def schedule(self):
        """ref: model.py:35-42: per-step lr for both groups, weight decay for group 0 only."""
        opt = self.trainer.optimizers[0]
        k = self._idx(self.global_step)
        for i, group in enumerate(opt.param_groups):
            group["lr"] = self.mylr_scheduler[k]
            if i == 0:
                group["weight_decay"] = self.wd_scheduler[k]
        if isinstance(opt, FusedHFAdamW):               # EMA of this step uses the post-increment index (Appendix A.6)
            opt.pending_ema = float(self.ema_scheduler[self._idx(self.global_step + 1)])
        self.log("wd", self.wd_scheduler[k], prog_bar=True, logger=True)
        self.log("lr", self.mylr_scheduler[k], prog_bar=True, logger=True)

    def configure_optimizers(self):
        """HF-AdamW semantics over the two reference groups (model.py:44-48), executed by the fused HIP kernel."""
        return [FusedHFAdamW(self.model, get_params_groups(self.model.student), lr=self.learning_rate, weight_decay=0.0)]

    def on_after_backward(self):
        self.model.engine.allreduce_grads()             # DDP mean; no-op at world size 1

    def on_train_batch_end(self, outputs, batch, batch_idx: int, unused: int = 0) -> None:
        """ref: model.py:49-51 (global_step has already been incremented)."""
        m = self.ema_scheduler[self._idx(self.global_step)]
        self.model.update_teacher(m)

    @staticmethod
    def add_model_specific_args(parent_parser):
        """ref: model.py:53-64."""
        parser = parent_parser.add_argument_group("ATSTModel")
        parser.add_argument("--arch", type=str, default="small")
        parser.add_argument("--learning_rate", default=0.0005, type=float,
                            help="peak learning rate after linear warm-up, for a reference batch size of 256")
        parser.add_argument("--ema", default=0.99, type=float, help="base EMA momentum, cosine-increased to 1")
        parser.add_argument("--warmup_steps", default=1300, type=int)
        parser.add_argument("--max_steps", default=39010, type=int)
        return parent_parser
