"""``FusedHFAdamW``: torch.optim.Optimizer facade over the fused HIP AdamW(+EMA) kernel.

Semantics = transformers.optimization.AdamW (<5.0) as configured by audiossl/methods/atst/model.py:44-48:
betas (0.9, 0.999), eps 1e-6, correct_bias, decoupled weight decay after the update, group 0 regularised / group 1 not.
The reference module overwrites ``param_groups[i]["lr"]`` and ``param_groups[0]["weight_decay"]`` every step; step()
reads them back from the groups, so ``ATSTLightningModule.schedule`` works unchanged."""
from __future__ import annotations

import torch


class FusedHFAdamW(torch.optim.Optimizer):
    def __init__(self, model, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.0):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._model = [model]
        self.pending_ema = None          # set by the module's schedule(); consumed by step()

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        model = self._model[0]
        g0 = self.param_groups[0]
        ema, self.pending_ema = self.pending_ema, None
        model.engine.optimizer_step(float(g0["lr"]), float(g0["weight_decay"]), ema, betas=g0["betas"], eps=g0["eps"])
        if ema is not None:
            model._ema_applied_step = model.engine.opt_step
        return loss

    def zero_grad(self, set_to_none: bool = True):
        # the engine zeroes its flat gradient buffer at the start of every backward; keep .grad views bound
        return None
