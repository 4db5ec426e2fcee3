"""``FrameATSTLightningModule`` -- audiossl/methods/atstframe/model.py:90-166 on the HIP engine (ATST-Frame recipe:
symmetric loss, Linear patch embed, pos_type 'cut').  Batch: ``((melspecs, lengths, masks), _)`` (model.py:120)."""
from __future__ import annotations

from ...models.atst import FrameATST
from ...utils.common import bool_flag
from ..atst.model import ATSTLightningModule


class FrameATSTLightningModule(ATSTLightningModule):
    def __init__(self, arch="small", learning_rate: float = 5e-4, warmup_steps=1300, max_steps=39000, ema=0.99, symmetric=True,
                 pos_type="cut", avg_blocks=0, patch_embed="Linear", **kwargs):
        self._frame_cfg = dict(symmetric=symmetric, pos_type=pos_type, avg_blocks=avg_blocks, patch_embed=patch_embed)
        super().__init__(arch=arch, learning_rate=learning_rate, warmup_steps=warmup_steps, max_steps=max_steps, ema=ema, **kwargs)
        self.symmetric = symmetric
        self.hparams.update(self._frame_cfg)

    def _build_model(self, arch, kwargs):
        # the reference hands **kwargs (= vars(args): spec_h = n_mels, patch_h, patch_w, ... train.py:15,17,50-51) down to FrameAST_small / _base,
        # which pick the geometry out of them (atstframe/audio_transformer.py:283-291); the same keys are picked here
        geo = {k: int(kwargs[k]) for k in ("patch_h", "patch_w", "spec_w") if kwargs.get(k) is not None}
        if kwargs.get("spec_h") is not None and int(kwargs["spec_h"]) != geo.get("patch_h", 64):
            raise NotImplementedError("one patch row: spec_h (= n_mels) must equal patch_h")
        return FrameATST(arch=arch, **self._frame_cfg, **geo)

    def training_step(self, batch, batch_idx):
        self.schedule()
        (melspecs, lengths, masks), _ = batch
        loss, std_frm_stu, std_frm_tea = self.model(melspecs, lengths, masks)
        self.log("loss", loss, prog_bar=True, logger=True)
        self.log("loss_frm", loss, prog_bar=True, logger=True)
        self.log("std_frm_tea", std_frm_tea, prog_bar=True, logger=True)
        self.log("std_frm_stu", std_frm_stu, prog_bar=True, logger=True)
        self.log("ema", self.ema_scheduler[self._idx(self.global_step)], prog_bar=True, logger=True)
        self.log("step", self.global_step, prog_bar=True, logger=True)
        return loss

    @staticmethod
    def add_model_specific_args(parent_parser):
        parser = parent_parser.add_argument_group("FrameATSTModel")
        parser.add_argument("--arch", type=str, default="small")
        parser.add_argument("--symmetric", type=bool_flag, default=True, help="whether or not using symmetric loss")
        parser.add_argument("--learning_rate", default=0.0005, type=float)
        parser.add_argument("--ema", default=0.99, type=float)
        parser.add_argument("--warmup_steps", default=1300, type=int)
        parser.add_argument("--max_steps", default=39010, type=int)
        parser.add_argument("--pos_type", type=str, default="cut")
        parser.add_argument("--avg_blocks", type=int, default=0)
        parser.add_argument("--patch_embed", type=str, default="Linear")
        return parent_parser
