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
    """Structural container (the reference's Block / Attention / Mlp / Sequential nodes hold parameters only here).
    Index access mirrors nn.ModuleList / nn.Sequential (``encoder.blocks[3]``, ``projector[0]``)."""

    def __getitem__(self, i):
        return self._modules[str(i)]

    def __len__(self):
        return len(self._modules)


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


class EncoderView(_Node):
    """``model.{student,teacher}.encoder``: parameter container + the reference's inference API, evaluated by the HIP
    encoder in eval semantics (no DropPath).  ref: audiossl/models/atst/audio_transformer.py:188-221 (forward),
    :235-255 (get_intermediate_layers), :257-366 (get_intermediate_layers_chunks / get_cls_avg)."""

    def _bind(self, engine, net):
        self._eng, self._netname = [engine], net
        self.embed_dim = engine.cfg["embed_dim"]
        self.use_cls = not engine.frame
        self.nprompt = 0

    @torch.no_grad()
    def _blocks(self, x, length, n):
        """-> (list of LN(x_i) [S, n_tok(+1), C] fp32 for the last n blocks, patch_length [S] (device), n_tok).
        Forward-only workspace (one layer of activations, re-used for every block) + fp32 taps of the last n block
        outputs; the per-geometry passes live in a small LRU so variable-length evaluation cannot grow HBM without bound."""
        from ... import hip
        eng = self._eng[0]
        eng.sync_shadows()
        x = x.to(eng.device, torch.float32).contiguous()
        S, width = x.shape[0], x.shape[-1]
        ep = eng.inference_pass(self._netname, S, width)
        if length is None:
            length = torch.full((S,), width, dtype=torch.int64)
        l = torch.as_tensor(length).to(torch.int64)
        pw = eng.patch_w
        plen = (l - l % pw) // pw                                 # reference patch_length, NOT clipped to the chunk
        valid = (torch.clamp(plen, max=ep.n_tok) + ep.use_cls).to(torch.int32)
        C = eng.cfg["embed_dim"]
        n = min(n, eng.depth)
        tap = torch.empty(n, ep.M, C, device=eng.device)
        ep.e.tap, ep.e.tap_first = tap.data_ptr(), eng.depth - n
        ep.forward(x, eng.upload(valid), None, None)
        ep.e.tap = None
        nf = "encoder.norm_frame" if eng.frame else "encoder.norm"
        gw, gb = eng.param_view(self._netname, nf + ".weight"), eng.param_view(self._netname, nf + ".bias")
        y = torch.empty(n, ep.M, C, device=eng.device)
        hip.call("atst_layernorm_fwd_f32", hip.ptr(tap), hip.ptr(gw), hip.ptr(gb), hip.ptr(y), n * ep.M, C, hip.stream())
        outs = [y[i].view(S, ep.RS, C)[:, :ep.n_tok + ep.use_cls] for i in range(n)]
        return outs, eng.upload(plen), ep.n_tok

    def forward(self, x, mask_index=None, length=None, avg=False):
        outs, plen, _ = self._blocks(x, length, 1)
        y = outs[0]
        if self.use_cls:
            return y[:, 0]
        m = (torch.arange(y.shape[1], device=y.device)[None, :] < plen[:, None]).unsqueeze(-1)
        return (y * m).sum(1) / plen[:, None]

    def get_intermediate_layers(self, x, length, n=1, scene=True):
        """Clip encoder: list of the last n normalised block outputs [S, 1 + T, C] (ref: audio_transformer.py:235-255).
        Frame encoder: FrameAST.get_intermediate_layers (ref: methods/atstframe/audio_transformer.py:259-281) -- the last n
        norm_frame'd block outputs concatenated on the feature axis, mean-pooled over the valid frames when ``scene``
        ([S, n*C]) or as frame sequences ([S, T, n*C]); consumed by atstframe/embedding.py:75,121."""
        outs, plen, _ = self._blocks(x, length, n)
        if self.use_cls:
            return outs
        if not scene:
            return torch.cat(outs, dim=-1)
        lm = (torch.arange(outs[0].shape[1], device=outs[0].device)[None, :] < plen[:, None]).unsqueeze(-1)
        return torch.cat([(o * lm).sum(1) / (plen[:, None] + 1e-6) for o in outs], dim=-1)

    def get_intermediate_layers_chunks(self, x, length, n=1, chunk_len=601, avgpool=True):
        total = x.shape[-1]
        length = torch.as_tensor(length)
        dev = self._eng[0].device
        cls_c, avg_c, marks = [], [], []
        for i in range(total // chunk_len + 1):
            start, end = i * chunk_len, min((i + 1) * chunk_len, total)
            if end <= start:
                continue
            cur = torch.clip(length - i * chunk_len, 0)
            mark = (cur > 0) if i == 0 else (cur > chunk_len // 2)
            outs, plen, _ = self._blocks(x[..., start:end], cur, n)
            off = 1 if self.use_cls else 0
            lm = (torch.arange(outs[0].shape[1] - off, device=outs[0].device)[None, :] < plen[:, None]).unsqueeze(-1)
            cls_c.append(torch.stack([o[:, 0] if self.use_cls else torch.zeros_like(o[:, 0]) for o in outs]))
            avg_c.append(torch.stack([(o[:, off:] * lm).sum(1) / (plen[:, None] + 1e-6) for o in outs]))
            marks.append(self._eng[0].upload(mark.float()) if not mark.is_cuda else mark.float())
        w = torch.stack(marks)[:, None, :, None]                        # [chunks, 1, S, 1]
        cls = (torch.stack(cls_c) * w).sum(0) / w.sum(0)                # [n, S, C]
        avg = (torch.stack(avg_c) * w).sum(0) / w.sum(0)
        parts = list(cls) + (list(avg) if avgpool else [])
        return torch.cat(parts, dim=-1)


class _Net(_Node):
    """MultiCropWrapper-shaped view of one network (encoder, projector[, predictor]). ref: byol.py:82-121."""

    def __init__(self, engine: AtstEngine, net: str):
        super().__init__()
        self._engine_ref = [engine]            # list: keep the engine out of the module tree
        self._net = net
        student = net == "student"
        self.add_module("encoder", EncoderView())
        self.encoder._bind(engine, net)
        for name in engine.layout.entries:
            if not student and name.startswith("predictor."):
                continue
            _attach(self, name, engine.param_view(net, name), requires_grad=student)
            if name.endswith(".1.bias"):        # BatchNorm1d buffers follow its parameters in state_dict order
                which = name.split(".")[0]
                for b, t in engine.bn_buffers[f"{net}.{which}"].items():
                    _attach(self, f"{which}.1.{b}", t, buffer=True)
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
        # patch geometry (the reference's patch_h / patch_w factory arguments, audio_transformer.py:367-374, with spec_h = n_mels = patch_h:
        # one patch row); pos_embed has spec_w // patch_w + 1 rows (audio_transformer.py:95-102)
        patch_h, patch_w = kwargs.pop("patch_h", 64), kwargs.pop("patch_w", 4)
        self.engine = AtstEngine(arch, frame=frame, depth=depth, ncrops=ncrops, drop_path_rate=drop,
                                 n_pos=spec_w // patch_w + 1, symmetric=kwargs.pop("symmetric", True),
                                 patch_embed=kwargs.pop("patch_embed", "Linear"), precise=kwargs.pop("precise", False),
                                 fp8=kwargs.pop("fp8", False), patch_h=patch_h, patch_w=patch_w)
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

    def load_state_dict(self, state_dict, strict=True, **kw):
        out = super().load_state_dict(state_dict, strict=strict, **kw)
        self.engine.sync_shadows(force=True)
        return out


class FrameATST(ATST):
    """ref: audiossl/methods/atstframe/model.py:24-85 (ATST-Frame branch: avg_blocks=0, Linear patch embed; symmetric=True: both
    views through both networks, cross-view loss; symmetric=False: teacher sees view 0, student the masked view 1, one pair)."""

    def __init__(self, arch="small", symmetric=True, pos_type="cut", avg_blocks=0, patch_embed="Linear", **kwargs):
        if pos_type != "cut" or avg_blocks != 0 or patch_embed not in ("Linear", "CNN"):
            raise NotImplementedError("HIP path implements the ATST-Frame branch of the reference: pos_type='cut', avg_blocks=0 "
                                      "(not the data2vec-style variant), patch_embed 'Linear' or 'CNN'; symmetric and asymmetric "
                                      "losses are both available")
        super().__init__(arch=arch, ncrops=2, frame=True, symmetric=symmetric, patch_embed=patch_embed, **kwargs)
        self.symmetric = bool(symmetric)

    def forward(self, x, length, mask, keep_teacher=None, keep_student=None):
        return super().forward(x, length, mask, keep_teacher, keep_student)
