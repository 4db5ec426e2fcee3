"""``ATST`` / ``FrameATST`` with the reference's module tree, state_dict keys and call signatures, backed by the HIP engine.

Mirrors audiossl/models/atst/atst.py:6-34 (ATST), audiossl/models/atst/byol.py:82-121 (MultiCropWrapper) and
audiossl/methods/atstframe/model.py:24-85 (FrameATST).  Parameters are views into the engine's flat fp32 buffers, so
``state_dict()`` / ``load_state_dict()`` round-trip with reference checkpoints, while compute never touches torch.nn.
"""
from __future__ import annotations

import torch
from torch import nn

from ...engine import AtstEngine


class _Node(nn.Module):
    """Structural container (the reference's Block / Attention / Mlp / Sequential nodes hold parameters only here)."""


def _attach(root: nn.Module, dotted: str, tensor: torch.Tensor, buffer: bool = False, requires_grad: bool = True):
    parts, m = dotted.split("."), root
    for p in parts[:-1]:
        if not hasattr(m, p):
            m.add_module(p, _Node())
        m = getattr(m, p)
    if buffer:
        m.register_buffer(parts[-1], tensor)
    else:
        m.register_parameter(parts[-1], nn.Parameter(tensor, requires_grad=requires_grad))


class _Net(_Node):
    """MultiCropWrapper-shaped view of one network (encoder, projector[, predictor]). ref: byol.py:82-121."""

    def __init__(self, engine: AtstEngine, net: str):
        super().__init__()
        self._engine_ref = [engine]            # list: keep the engine out of the module tree
        self._net = net
        student = net == "student"
        for name in engine.layout.entries:
            if not student and name.startswith("predictor."):
                continue
            _attach(self, name, engine.param_view(net, name), requires_grad=student)
            if name.endswith(".1.bias"):        # BatchNorm1d buffers follow its parameters in state_dict order
                which = name.split(".")[0]
                for b, t in engine.bn_buffers[f"{net}.{which}"].items():
                    _attach(self, f"{which}.1.{b}", t, buffer=True)
        self.encoder.embed_dim = engine.cfg["embed_dim"]
        if not student:
            self.predictor = nn.Identity()

    def forward(self, x, length, avg=False):
        raise RuntimeError("call ATST.forward / FrameATST.forward: student and teacher passes are fused in the HIP engine")


class _StepFn(torch.autograd.Function):
    """loss = engine.forward(...) ; backward runs the HIP backward into the flat gradient buffer and exposes it as
    ``param.grad`` views, so ``loss.backward()`` + any torch optimizer (or the fused one) work unchanged."""

    @staticmethod
    def forward(ctx, anchor, model, mels, lengths, masks, keep_t, keep_s, train):
        loss, std_s, std_t = model.engine.forward(mels, lengths, masks, keep_t, keep_s, train=train)
        ctx.model = model
        return loss.reshape(()).clone(), std_s.detach(), std_t.detach()

    @staticmethod
    def backward(ctx, g_loss, g_s, g_t):
        model = ctx.model
        model.engine.backward(grad_scale=g_loss)
        model._bind_grads()
        return (torch.zeros_like(g_loss),) + (None,) * 7


class ATST(nn.Module):
    """ref: audiossl/models/atst/atst.py:6-34.  ``frame=True`` gives FrameATST (symmetric, Linear patch embed)."""

    def __init__(self, arch="small", ncrops=2, frame=False, **kwargs):
        super().__init__()
        drop = kwargs.pop("drop_path_rate", 0.1)
        depth = kwargs.pop("depth", None)
        spec_w = kwargs.pop("spec_w", 1001)
        self.engine = AtstEngine(arch, frame=frame, depth=depth, ncrops=ncrops, drop_path_rate=drop,
                                 n_pos=spec_w // 4 + 1)
        self.ncrops, self.frame = ncrops, frame
        self.engine.init_weights()
        self.student = _Net(self.engine, "student")
        self.teacher = _Net(self.engine, "teacher")
        self._anchor = torch.zeros((), device=self.engine.device, requires_grad=True)
        self._ema_applied_step = -1

    # -- reference API ----------------------------------------------------------------------------------------------
    def forward(self, melspecs, lengths, masks=None, keep_teacher=None, keep_student=None):
        """-> (loss, std_cls_s, std_cls_t).  ref: atst.py:24-28 / atstframe/model.py:68-76."""
        return _StepFn.apply(self._anchor, self, list(melspecs), list(lengths), None if masks is None else list(masks),
                             keep_teacher, keep_student, torch.is_grad_enabled())

    def update_teacher(self, m):
        """EMA over encoder + projector parameters (BN buffers excluded). ref: atst.py:29-34."""
        if self._ema_applied_step == self.engine.opt_step and self.engine.opt_step > 0:
            return                                  # already fused into the optimizer kernel for this step
        self.engine.ema_update(float(m))

    # -- plumbing ---------------------------------------------------------------------------------------------------
    def _bind_grads(self):
        eng = self.engine
        for name, p in self.student.named_parameters():
            if name == "encoder.mask_embed" and not self.frame:
                p.grad = None                       # never used in clip-level ATST (reference: grad is None)
            else:
                p.grad = eng.param_view("student", name, grad=True)

    def _load_from_state_dict(self, *args, **kwargs):
        super()._load_from_state_dict(*args, **kwargs)

    def load_state_dict(self, state_dict, strict=True, **kw):
        out = super().load_state_dict(state_dict, strict=strict, **kw)
        self.engine.sync_shadows(force=True)
        return out


class FrameATST(ATST):
    """ref: audiossl/methods/atstframe/model.py:24-85 (ATST-Frame branch: avg_blocks=0, symmetric, Linear patch embed)."""

    def __init__(self, arch="small", symmetric=True, pos_type="cut", avg_blocks=0, patch_embed="Linear", **kwargs):
        if not symmetric or pos_type != "cut" or avg_blocks != 0 or patch_embed != "Linear":
            raise NotImplementedError("HIP path implements the shipped ATST-Frame recipe: symmetric, pos_type='cut', "
                                      "avg_blocks=0, patch_embed='Linear' (methods/atstframe/train_small.sh)")
        super().__init__(arch=arch, ncrops=2, frame=True, **kwargs)
        self.symmetric = True

    def forward(self, x, length, mask, keep_teacher=None, keep_student=None):
        return super().forward(x, length, mask, keep_teacher, keep_student)
