"""Host-side driver of the HIP hot path: flat parameter buffers, encoder passes, BYOL heads, loss, fused optimizer.

Mirrors what ``ATST.forward`` + autograd + ``AdamW.step`` + ``ATST.update_teacher`` do in the reference
(audiossl/models/atst/atst.py:24-34, audiossl/models/atst/byol.py:57-121, audiossl/methods/atst/model.py:24-51 and the
ATST-Frame twins under audiossl/methods/atstframe/), but every tensor op is a call into libatst_hip.so.
torch is used for device memory, streams and torch.distributed only.
"""
from __future__ import annotations

import ctypes as C
import math
import os
import warnings
from collections import OrderedDict
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist

from . import hip, parallel

ARCH = {"small": dict(embed_dim=384, depth=12, num_heads=6), "base": dict(embed_dim=768, depth=12, num_heads=12)}
ALIGN = 256          # every tensor starts on a 256-element boundary of the flat buffers (one flag byte per chunk)
BN_EPS, BN_MOMENTUM = 1e-5, 0.1
HEAD_HIDDEN, HEAD_OUT = 4096, 256


def _round_up(n, a=ALIGN):
    return (n + a - 1) // a * a


def encoder_param_shapes(arch: str, depth: Optional[int] = None, frame: bool = False, n_pos: int = 251, patch_embed: str = "Linear",
                         patch_h: int = 64, patch_w: int = 4):
    """Parameter names/shapes in the reference's registration order (= state_dict order).
    ref: audiossl/models/atst/audio_transformer.py:80-120 ; audiossl/methods/atstframe/audio_transformer.py:101-149."""
    cfg = ARCH[arch]
    d, depth = cfg["embed_dim"], (cfg["depth"] if depth is None else depth)
    out = [("mask_embed", (1, 1, d))]
    if not frame:
        out.append(("cls_token", (1, 1, d)))
    # patch_embed="CNN" (ATST-Frame option, atstframe/audio_transformer.py:57-74,117-118): Conv2d(1, d, (64, 4), stride (64, 4)) -- the
    # same contraction over k = f * 4 + t, stored as [d, 1, 64, 4] under `patch_embed.proj.*`
    pe = [("patch_embed.proj.weight", (d, 1, patch_h, patch_w)), ("patch_embed.proj.bias", (d,))] if patch_embed == "CNN" else \
         [("patch_embed.patch_embed.weight", (d, patch_h * patch_w)), ("patch_embed.patch_embed.bias", (d,))]
    out += [("pos_embed", (1, n_pos, d))] + pe
    for i in range(depth):
        b = f"blocks.{i}."
        out += [(b + "norm1.weight", (d,)), (b + "norm1.bias", (d,)), (b + "attn.qkv.weight", (3 * d, d)),
                (b + "attn.proj.weight", (d, d)), (b + "attn.proj.bias", (d,)), (b + "norm2.weight", (d,)),
                (b + "norm2.bias", (d,)), (b + "mlp.fc1.weight", (4 * d, d)), (b + "mlp.fc1.bias", (4 * d,)),
                (b + "mlp.fc2.weight", (d, 4 * d)), (b + "mlp.fc2.bias", (d,))]
    nf = "norm_frame" if frame else "norm"
    out += [(nf + ".weight", (d,)), (nf + ".bias", (d,))]
    return out


def head_param_shapes(in_dim: int):
    """ref: audiossl/models/atst/byol.py:6-22 (Linear no-bias, BatchNorm1d affine, ReLU, Linear no-bias)."""
    return [("0.weight", (HEAD_HIDDEN, in_dim)), ("1.weight", (HEAD_HIDDEN,)), ("1.bias", (HEAD_HIDDEN,)),
            ("3.weight", (HEAD_OUT, HEAD_HIDDEN))]


class FlatLayout:
    """name -> (offset, shape) for 'encoder.*', 'projector.*', 'predictor.*' in one flat buffer."""

    def __init__(self, arch: str, depth: Optional[int], frame: bool, patch_embed: str = "Linear", n_pos: int = 251,
                 patch_h: int = 64, patch_w: int = 4):
        d = ARCH[arch]["embed_dim"]
        self.entries: "OrderedDict[str, Tuple[int, Tuple[int, ...]]]" = OrderedDict()
        off = 0
        groups = [("encoder.", encoder_param_shapes(arch, depth, frame, n_pos, patch_embed, patch_h, patch_w)), ("projector.", head_param_shapes(d)),
                  ("predictor.", head_param_shapes(HEAD_OUT))]
        for prefix, shapes in groups:
            for name, shape in shapes:
                self.entries[prefix + name] = (off, shape)
                off = _round_up(off + math.prod(shape))
            if prefix == "projector.":
                self.n_teacher = off
        self.n_student = off

    def numel(self, name):
        return math.prod(self.entries[name][1])


class Workspace:
    """A caller-owned device byte buffer carved by the C engine; exposes typed torch views at raw pointers."""

    def __init__(self, nbytes: int, device):
        self.buf = torch.empty(nbytes, dtype=torch.uint8, device=device)

    def view(self, ptr: int, shape, dtype):
        off = ptr - self.buf.data_ptr()
        n = math.prod(shape) * torch.empty(0, dtype=dtype).element_size()
        assert 0 <= off and off + n <= self.buf.numel()
        return self.buf[off:off + n].view(dtype).view(*shape)


class Uploader:
    """Host -> device copies of the small per-step index / length / mask tensors WITHOUT a stream synchronisation.
    torch's blocking copy from pageable memory ends in hipStreamSynchronize (drains the launch queue once per call);
    here the bytes are staged in a ring of pinned slots and copied with non_blocking=True.  A slot is re-used only after
    the event recorded behind its last copy has completed (host-side wait on that event only, never a stream drain)."""

    def __init__(self, device, slot_bytes: int = 1 << 20, slots: int = 32):
        self.device, self.slot_bytes, self.slots = device, slot_bytes, slots
        self.buf = torch.empty(slots, slot_bytes, dtype=torch.uint8).pin_memory()
        self.buf_np = self.buf.numpy()                              # the same pinned bytes: staged with a numpy copy (no intra-op thread pool, see AtstEngine._frame_rows)
        self.events: List[Optional[torch.cuda.Event]] = [None] * slots
        self.k = 0

    def __call__(self, t: torch.Tensor) -> torch.Tensor:
        t = t.contiguous()
        if t.is_cuda:
            return t
        n = t.numel() * t.element_size()
        if n == 0:
            return torch.empty(t.shape, dtype=t.dtype, device=self.device)
        if n > self.slot_bytes:                                     # rare (large mels arrive on the device already)
            return t.pin_memory().to(self.device, non_blocking=True)
        i = self.k % self.slots
        self.k += 1
        if self.events[i] is not None:
            self.events[i].synchronize()
        stage = self.buf[i, :n].view(t.dtype).view(t.shape)
        try:
            self.buf_np[i, :n] = t.numpy().reshape(-1).view(np.uint8)
        except (TypeError, ValueError, RuntimeError):               # dtypes numpy does not know (bf16): torch's copy
            stage.copy_(t)
        d = stage.to(self.device, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.events[i] = ev
        return d


def _host_cat(ts):
    """torch.cat for the small per-step host tensors (lengths, masks): numpy when they all live on the host (no intra-op thread pool: see _frame_rows)."""
    if len(ts) == 1:
        return ts[0]
    if all(not t.is_cuda for t in ts):
        return torch.from_numpy(np.concatenate([t.numpy() for t in ts]))
    return torch.cat(ts)


def pad_tokens(n: int) -> int:
    for np_ in (32, 64, 128, 256):
        if n <= np_:
            return np_
    raise hip.HipError(f"{n} tokens per sequence exceed the 256-token attention kernels (10 s @ 16 kHz = 251)")


class EncoderPass:
    """One encoder invocation geometry (S sequences of one mel width) with its activation workspace."""

    def __init__(self, eng: "AtstEngine", net: str, S: int, width: int, train: bool, precise: bool = False, record_amax: bool = False):
        self.eng, self.net, self.S, self.width, self.train = eng, net, S, width, train
        self.precise = bool(precise)                     # fp32 / split-bf16 twin of the encoder (csrc/engine_hp.hip): parity mode
        cfg = eng.cfg
        pw = eng.patch_w
        self.n_tok = (width - width % pw) // pw
        self.use_cls = 0 if eng.frame else 1
        if self.n_tok + 1 > eng.n_pos:                          # the table kernel reads pos row n+1 for patch n in BOTH modes
            raise hip.HipError(f"mel width {width} needs {self.n_tok + 1} positions; pos_embed has {eng.n_pos} "
                               "(pos_type='cut', ref: audio_transformer.py:95-102)")
        self.NP = pad_tokens(self.n_tok + self.use_cls)
        # Row stride between sequences.  Short views are PACKED (1 s: 26 tokens in attention tiles of 32 -> 19 % fewer rows in every
        # GEMM / LayerNorm / weight-gradient launch of the group) when the packed row count keeps the 64-row alignment the tall
        # weight-gradient tile wants; the 256-token kernels and the parity-mode twin keep the padded layout.  ATST_PACK=0 disables it.
        self.RS = self.NP
        toks = self.n_tok + self.use_cls
        if not self.precise and self.NP < 256 and toks < self.NP and (S * toks) % 64 == 0 and os.environ.get("ATST_PACK", "1") != "0":
            self.RS = toks
        self.M = S * self.RS
        lib = hip.load()
        if self.precise:
            nbytes = lib.atst_encoder_hp_ws_bytes(S, self.NP, cfg["embed_dim"], cfg["num_heads"], eng.depth, eng.patch_h, eng.patch_w)
        else:
            nbytes = lib.atst_encoder_ws_bytes_geo(S, self.NP, cfg["embed_dim"], cfg["num_heads"], eng.depth, int(train), int(eng.fp8),
                                                   eng.patch_h, eng.patch_w)
        self.ws = Workspace(nbytes, eng.device)
        e = hip.Encoder()
        e.S, e.NP, e.n_tok, e.width, e.C, e.H, e.depth = S, self.NP, self.n_tok, width, cfg["embed_dim"], cfg["num_heads"], eng.depth
        e.use_cls, e.train = self.use_cls, int(train)
        e.patch_h, e.patch_w = eng.patch_h, eng.patch_w
        e.row_stride = self.RS
        if net == "student":
            e.p32, e.p16, e.p16t, e.g32 = eng.p32.data_ptr(), eng.p16.data_ptr(), eng.p16t.data_ptr(), eng.g32.data_ptr()
        else:
            e.p32, e.p16, e.p16t, e.g32 = eng.t32.data_ptr(), eng.t16.data_ptr(), None, None
        e.off = eng.enc_off
        if eng.fp8 and not self.precise:
            e.fp8 = 1
            e.p8, e.w_dq = (eng.p8.data_ptr(), eng.dq_s.data_ptr()) if net == "student" else (eng.t8.data_ptr(), eng.dq_t.data_ptr())
            k = 0 if net == "student" else 1
            e.f8_sat = eng.f8_sat[k:].data_ptr()
            # running activation scales of the e4m3 forward (delayed scaling): every pass quantises with them; the passes of a training
            # step also record this step's amax (inference passes do not touch the state)
            e.f8_act_scale = eng.f8a_scale[k].data_ptr()
            if record_amax:
                e.f8_act_amax = eng.f8a_amax_sites[k].data_ptr()
            if net == "student" and train:
                e.p8t, e.g8_scale, e.g8_amax = eng.p8t.data_ptr(), eng.g8_scale.data_ptr(), eng.g8_amax_sites.data_ptr()
                # e4m3 weight gradients (fc1 / fc2 / proj): the backward needs the activation scales the forward of THIS step quantised with --
                # f8a_scale itself is advanced by _fp8_after_forward() between the two
                e.fp8_wgrad = eng.fp8_wgrad_mode()
                e.f8_act_scale_bwd = eng.f8a_scale_used.data_ptr()
        e.ws, e.ws_bytes = self.ws.buf.data_ptr(), nbytes
        self.e = e
        if self.precise:
            self.out = self.ws.view(lib.atst_encoder_hp_out(C.byref(e)), (self.M, e.C), torch.float32)
            self.dout = self.ws.view(lib.atst_encoder_hp_dout(C.byref(e)), (self.M, e.C), torch.float32)
        else:
            self.out = self.ws.view(lib.atst_encoder_out(C.byref(e)), (self.M, e.C), torch.bfloat16)
            self.dout = self.ws.view(lib.atst_encoder_dout(C.byref(e)), (self.M, e.C), torch.bfloat16) if train else None
        self._keep = None

    def forward(self, mel: torch.Tensor, valid: torch.Tensor, rowflag: Optional[torch.Tensor], dp_scale: Optional[torch.Tensor]):
        assert mel.shape == (self.S, 1, self.eng.patch_h, self.width) and mel.dtype == torch.float32
        valid = self.eng.upload(valid)                       # host-side lengths: pinned staging, no stream sync
        self._keep = (mel, valid, rowflag, dp_scale)        # keep inputs alive until backward
        e = self.e
        e.mel, e.valid = hip.ptr(mel), hip.ptr(valid)
        e.rowflag, e.dp_scale = hip.ptr(rowflag), hip.ptr(dp_scale)
        e.fp8_lean = self._lean_mode()                      # bf16 activation copies the announced backward will not read are not written
        if getattr(e, "fp8", 0) and self.net == "student" and self.train and self.eng.f8a_scale_used is not None:
            # The e4m3 weight gradients of this pass de-scale with the activation scales THIS forward quantises with; the running scales move on
            # between forward and backward.  The snapshot is taken here, per pass and in stream order in front of the forward (every student pass
            # of a step copies the same values), so a caller that drives EncoderPass.forward() / backward() directly cannot pair a forward with a
            # stale snapshot (it used to be refreshed only by AtstEngine._fp8_after_forward()).
            self.eng.f8a_scale_used.copy_(self.eng.f8a_scale[0])
        if self.precise:
            hip.check(hip.load().atst_encoder_hp_fwd(C.byref(e), hip.stream()), "atst_encoder_hp_fwd")
        else:
            hip.check(hip.load().atst_encoder_fwd(C.byref(e), hip.stream()), "atst_encoder_fwd")
        return self.out

    def _lean_mode(self) -> int:
        """atst_encoder_t.fp8_lean of a training pass: 1 when its backward will run the e4m3 fc1 / fc2 / proj weight gradients, 2 when the qkv one too."""
        eng, e = self.eng, self.e
        if not (eng.fp8 and e.train and getattr(eng, "fp8_lean", False)) or getattr(eng, "fp8_bwd_state", 0) != 2 or self.M % 64:
            return 0
        mode = eng.fp8_wgrad_mode()
        return 0 if mode == 0 else (2 if (mode == 2 and hip.load().atst_attention_fp8_ok(int(e.NP), int(e.H), 1)) else 1)   # NP = 256 and (round 6) 32: the attention backward writes dqkv as e4m3

    def _check_lean(self):
        if self.e.fp8_lean and self.e.fp8_lean > self._lean_mode():
            raise RuntimeError("fp8: the forward of this pass left out bf16 activation copies (fp8_lean) that the backward now asked for would read -- "
                               "fp8_bwd_state / fp8_wgrad / fp8_qkv_state must not change between a forward and its backward")

    def backward(self):
        self._check_lean()
        self.e.fp8_bwd = int(getattr(self.eng, "fp8_bwd_state", 0)) if self.eng.fp8 else 0
        self.e.fp8_wgrad = self.eng.fp8_wgrad_mode()
        if self.precise:
            hip.check(hip.load().atst_encoder_hp_bwd(C.byref(self.e), hip.stream()), "atst_encoder_hp_bwd")
        else:
            hip.check(hip.load().atst_encoder_bwd(C.byref(self.e), hip.stream()), "atst_encoder_bwd")

    def backward_part(self, part: int, split: int):
        """part 0: final LayerNorm + blocks [split, depth) ; part 1: blocks [0, split) + token stage."""
        self._check_lean()
        self.e.fp8_bwd = int(getattr(self.eng, "fp8_bwd_state", 0)) if self.eng.fp8 else 0
        self.e.fp8_wgrad = self.eng.fp8_wgrad_mode()
        hip.check(hip.load().atst_encoder_bwd_part(C.byref(self.e), part, split, hip.stream()), "atst_encoder_bwd_part")

    def backward_range(self, lo: int, hi: int):
        """blocks [lo, hi) descending (+ final LayerNorm when hi == depth, + token stage when lo == 0)."""
        self._check_lean()
        self.e.fp8_bwd = int(getattr(self.eng, "fp8_bwd_state", 0)) if self.eng.fp8 else 0
        self.e.fp8_wgrad = self.eng.fp8_wgrad_mode()
        hip.check(hip.load().atst_encoder_bwd_range(C.byref(self.e), lo, hi, hip.stream()), "atst_encoder_bwd_range")

    def tokens(self):
        return self.ws.view(hip.load().atst_encoder_tokens(C.byref(self.e)), (self.M, self.e.C), torch.float32)

    def block_out(self, i):
        f = hip.load().atst_encoder_hp_block_out if self.precise else hip.load().atst_encoder_block_out
        return self.ws.view(f(C.byref(self.e), i), (self.M, self.e.C), torch.float32)


def _rows_buf(R: int, cols: int, dtype, dev) -> torch.Tensor:
    """[R, cols] buffer whose ALLOCATION does not depend on the exact row count: ATST-Frame's head batches are the masked rows of the step (about
    83 k at 256 clips, a different count every step), and torch's caching allocator answers GB-sized requests of ever-changing size with fresh
    hipMalloc / hipFree calls (a device-wide synchronisation each) -- measured as 1.1 - 5x slower Frame steps on some boxes and not on others
    (round 5).  The capacity is the row count rounded up to 4096 rows; the [:R] view is contiguous."""
    cap = R if R <= 4096 else (R + 4095) // 4096 * 4096
    return torch.empty(cap, cols, dtype=dtype, device=dev)[:R]


class HeadPass:
    """Linear(no bias) -> BatchNorm1d(train, cross-rank statistics) -> ReLU -> Linear(no bias), forward and backward.
    ref: audiossl/models/atst/byol.py:6-22 ; SyncBatchNorm semantics from Trainer(sync_batchnorm=True), methods/atst/train.py:22."""

    def __init__(self, eng: "AtstEngine", net: str, which: str, in_dim: int):
        self.eng, self.net, self.which, self.in_dim = eng, net, which, in_dim
        self.saved = None
        self._pending = None

    def _w(self, name, transposed=False, f32=False, grad=False):
        eng = self.eng
        off, shape = eng.layout.entries[f"{self.which}.{name}"]
        n = math.prod(shape)
        if grad:
            return eng.g32[off:off + n]
        if f32:
            return (eng.p32 if self.net == "student" else eng.t32)[off:off + n]
        if transposed:
            return eng.p16t[off:off + n]
        return (eng.p16 if self.net == "student" else eng.t16)[off:off + n]

    def forward(self, x: torch.Tensor, train: bool) -> torch.Tensor:
        """One head on its own: local statistics, cross-rank combine, BatchNorm + ReLU + second Linear."""
        local = self.forward_stats(x)
        return self.forward_finish(*parallel.combine_bn_stats(*local), train)

    def forward_stats(self, x: torch.Tensor):
        """First Linear + LOCAL BatchNorm statistics (mean, M2, row count).  Split from forward_finish() so that heads whose inputs
        do not depend on each other (teacher projector, student projector) share ONE cross-rank exchange."""
        eng, R = self.eng, x.shape[0]
        dev, st = x.device, hip.stream()
        # split-bf16 operands ([hi|lo|hi] x [hi|hi|lo] along K): the Linear in front of BatchNorm+ReLU is evaluated to
        # ~2^-16 so that bf16 noise does not flip ReLU gates (DESIGN.md "Precision")
        K = self.in_dim
        x3 = _rows_buf(R, 3 * K, torch.bfloat16, dev)
        hip.call("atst_split3_bf16", hip.ptr(x), R, K, 0, hip.ptr(x3), st)
        w3 = torch.empty(HEAD_HIDDEN, 3 * K, dtype=torch.bfloat16, device=dev)
        hip.call("atst_split3_bf16", hip.ptr(self._w("0.weight", f32=True)), HEAD_HIDDEN, K, 1, hip.ptr(w3), st)
        h = _rows_buf(R, HEAD_HIDDEN, torch.float32, dev)
        _gemm(x3, w3, R, HEAD_HIDDEN, 3 * K, hip.EPI_F32, h)
        x16 = x3                                            # columns [0,K) = bf16(x): wgrad operand, ld = 3K
        mean, m2 = torch.empty(HEAD_HIDDEN, device=dev), torch.empty(HEAD_HIDDEN, device=dev)
        scratch = torch.empty(32 * HEAD_HIDDEN, device=dev)                          # row-block partials (fixed-order reduction)
        hip.call("atst_bn_stats_f32", hip.ptr(h), R, HEAD_HIDDEN, hip.ptr(mean), hip.ptr(m2), hip.ptr(scratch), st)
        self._pending = (x16, h)
        return mean, m2, float(R)

    def forward_finish(self, mean, m2, count, train: bool) -> torch.Tensor:
        """SyncBatchNorm with the GLOBAL statistics (count-weighted combine over ranks; ragged per-rank row counts of ATST-Frame are
        handled; count: python float at world size 1, 0-dim device tensor across ranks, never read back) -> ReLU -> second Linear."""
        eng = self.eng
        x16, h = self._pending
        self._pending = None
        R, dev, st = h.shape[0], h.device, hip.stream()
        bn = eng.bn_buffers[f"{self.net}.{self.which}"]
        rstd = torch.empty(HEAD_HIDDEN, device=dev)
        on_dev = isinstance(count, torch.Tensor)
        cdev = count.to(torch.float32).contiguous() if on_dev else None        # named: must outlive the launch call
        # rstd + running statistics + num_batches_tracked in one launch (nn.BatchNorm1d(train), models/atst/byol.py:13-16)
        hip.call("atst_bn_finish_f32", hip.ptr(mean), hip.ptr(m2), 0.0 if on_dev else float(count), hip.ptr(cdev), BN_MOMENTUM, BN_EPS,
                 hip.ptr(bn["running_mean"]), hip.ptr(bn["running_var"]), hip.ptr(bn["num_batches_tracked"]), hip.ptr(rstd), HEAD_HIDDEN, st)
        out = _rows_buf(R, HEAD_OUT, torch.float32, dev)
        if eng.precise or eng.head_split2 == "all" or (eng.head_split2 == "gated" and self.net == "student" and self.which == "projector"):
            # second Linear in split-bf16 where its output feeds ANOTHER head's BatchNorm+ReLU gates: the student projector (the predictor follows)
            y3 = _rows_buf(R, 3 * HEAD_HIDDEN, torch.bfloat16, dev)
            hip.call("atst_bn_apply_relu_split3_bf16", hip.ptr(h), hip.ptr(mean), hip.ptr(rstd), hip.ptr(self._w("1.weight", f32=True)),
                     hip.ptr(self._w("1.bias", f32=True)), R, HEAD_HIDDEN, hip.ptr(y3), st)
            w3b = torch.empty(HEAD_OUT, 3 * HEAD_HIDDEN, dtype=torch.bfloat16, device=dev)
            hip.call("atst_split3_bf16", hip.ptr(self._w("3.weight", f32=True)), HEAD_OUT, HEAD_HIDDEN, 1, hip.ptr(w3b), st)
            _gemm(y3, w3b, R, HEAD_OUT, 3 * HEAD_HIDDEN, hip.EPI_F32, out)
            y16 = y3                                        # columns [0, 4096) = bf16(y), ld = 3 * 4096
        else:
            # ... and on plain bf16 operands where it feeds the loss only (teacher projector, student predictor): a third of the activation bytes
            # and of the second GEMM's contraction (ATST-Frame: ~83 k rows x 4096)
            y16 = _rows_buf(R, HEAD_HIDDEN, torch.bfloat16, dev)
            hip.call("atst_bn_apply_relu_bf16", hip.ptr(h), hip.ptr(mean), hip.ptr(rstd), hip.ptr(self._w("1.weight", f32=True)),
                     hip.ptr(self._w("1.bias", f32=True)), R, HEAD_HIDDEN, hip.ptr(y16), st)
            _gemm(y16, self._w("3.weight"), R, HEAD_OUT, HEAD_HIDDEN, hip.EPI_F32, out)
        if train:
            self.saved = (x16, h, mean, rstd, y16, count)
        return out

    def _backward_precise(self, dout: torch.Tensor) -> torch.Tensor:
        """Parity mode (AtstEngine(precise=True)): the same backward with split-bf16 operands everywhere -- dgrad operands split
        along the contraction axis ([hi|lo|hi] x [hi|hi|lo]), weight-gradient operands split by rows ([hi;lo;hi] x [hi;hi;lo]) --
        and an fp32 dh.  Same kernels (MFMA GEMMs, BatchNorm sums), ~2^-17 instead of 2^-9 per operand."""
        eng = self.eng
        x3, h, mean, rstd, y3, count = self.saved
        R, dev, st, K = x3.shape[0], x3.device, hip.stream(), self.in_dim

        def split_cols(x, rows, cols, b_layout):                       # fp32 [rows, cols] -> bf16 [rows, 3 cols]
            out = _rows_buf(rows, 3 * cols, torch.bfloat16, dev)
            hip.call("atst_split3_bf16", hip.ptr(x.contiguous()), rows, cols, b_layout, hip.ptr(out), st)
            return out

        def rows_dy(x):                                                 # fp32 [R, n] -> bf16 [3 R, n] = [hi; lo; hi]
            hi = x.to(torch.bfloat16)
            lo = (x - hi.float()).to(torch.bfloat16)
            return torch.cat([hi, lo, hi]).contiguous()

        def rows_x(x3_, n):                                             # saved [R, 3 n] = [hi | lo | hi] -> bf16 [3 R, n] = [hi; hi; lo]
            hi, lo = x3_[:, :n], x3_[:, n:2 * n]
            return torch.cat([hi, hi, lo]).contiguous()

        w3 = self._w("3.weight", f32=True).view(HEAD_OUT, HEAD_HIDDEN)
        w0 = self._w("0.weight", f32=True).view(HEAD_HIDDEN, K)
        _wgrad(rows_dy(dout), rows_x(y3, HEAD_HIDDEN), 3 * R, HEAD_OUT, HEAD_HIDDEN, self._w("3.weight", grad=True))
        dy = _rows_buf(R, HEAD_HIDDEN, torch.float32, dev)
        _gemm(split_cols(dout, R, HEAD_OUT, 0), split_cols(w3.t(), HEAD_HIDDEN, HEAD_OUT, 1), R, HEAD_HIDDEN, 3 * HEAD_OUT, hip.EPI_F32, dy)
        gamma, beta = self._w("1.weight", f32=True), self._w("1.bias", f32=True)
        s1, s2 = torch.empty(HEAD_HIDDEN, device=dev), torch.empty(HEAD_HIDDEN, device=dev)
        scratch = torch.empty(2 * 32 * HEAD_HIDDEN, device=dev)
        hip.call("atst_bn_relu_bwd_sums", hip.ptr(dy), hip.ptr(h), hip.ptr(mean), hip.ptr(rstd), hip.ptr(gamma), hip.ptr(beta),
                 R, HEAD_HIDDEN, hip.ptr(s1), hip.ptr(s2), hip.ptr(scratch), st)
        self._w("1.bias", grad=True).add_(s1)
        self._w("1.weight", grad=True).add_(s2)
        s1, s2 = parallel.allreduce_bn_backward_sums(s1, s2)
        inv = 1.0
        if isinstance(count, torch.Tensor):
            s1, s2 = (s1 / count).contiguous(), (s2 / count).contiguous()
        else:
            inv = 1.0 / count
        dh = _rows_buf(R, HEAD_HIDDEN, torch.float32, dev)
        hip.call("atst_bn_bwd_dx_f32", hip.ptr(dy), hip.ptr(h), hip.ptr(mean), hip.ptr(rstd), hip.ptr(gamma), hip.ptr(beta),
                 hip.ptr(s1), hip.ptr(s2), inv, R, HEAD_HIDDEN, hip.ptr(dh), st)
        _wgrad(rows_dy(dh), rows_x(x3, K), 3 * R, HEAD_HIDDEN, K, self._w("0.weight", grad=True))
        dx = _rows_buf(R, K, torch.float32, dev)
        _gemm(split_cols(dh, R, HEAD_HIDDEN, 0), split_cols(w0.t(), K, HEAD_HIDDEN, 1), R, K, 3 * HEAD_HIDDEN, hip.EPI_F32, dx)
        self.saved = None
        return dx

    def backward(self, dout: torch.Tensor) -> torch.Tensor:
        eng = self.eng
        if eng.precise:
            return self._backward_precise(dout)
        x16, h, mean, rstd, y16, count = self.saved
        R, dev, st = x16.shape[0], x16.device, hip.stream()
        d16 = _rows_buf(R, HEAD_OUT, torch.bfloat16, dev)
        hip.call("atst_cast_bf16", hip.ptr(dout), R * HEAD_OUT, hip.ptr(d16), st)
        _wgrad(d16, y16, R, HEAD_OUT, HEAD_HIDDEN, self._w("3.weight", grad=True), ldx=y16.shape[1])       # [hi | lo | hi] or plain bf16(y)
        dy = _rows_buf(R, HEAD_HIDDEN, torch.float32, dev)
        _gemm(d16, self._w("3.weight", transposed=True), R, HEAD_HIDDEN, HEAD_OUT, hip.EPI_F32, dy)
        gamma, beta = self._w("1.weight", f32=True), self._w("1.bias", f32=True)
        s1, s2 = torch.empty(HEAD_HIDDEN, device=dev), torch.empty(HEAD_HIDDEN, device=dev)
        scratch = torch.empty(2 * 32 * HEAD_HIDDEN, device=dev)                      # row-block partials (fixed-order reduction)
        hip.call("atst_bn_relu_bwd_sums", hip.ptr(dy), hip.ptr(h), hip.ptr(mean), hip.ptr(rstd), hip.ptr(gamma), hip.ptr(beta),
                 R, HEAD_HIDDEN, hip.ptr(s1), hip.ptr(s2), hip.ptr(scratch), st)
        self._w("1.bias", grad=True).add_(s1)          # local sums: DDP averages parameter gradients afterwards
        self._w("1.weight", grad=True).add_(s2)
        s1, s2 = parallel.allreduce_bn_backward_sums(s1, s2)
        inv = 1.0
        if isinstance(count, torch.Tensor):                 # global row count lives on the device: fold 1/count into the sums
            s1, s2 = (s1 / count).contiguous(), (s2 / count).contiguous()
        else:
            inv = 1.0 / count
        dh16 = _rows_buf(R, HEAD_HIDDEN, torch.bfloat16, dev)
        hip.call("atst_bn_bwd_dx_bf16", hip.ptr(dy), hip.ptr(h), hip.ptr(mean), hip.ptr(rstd), hip.ptr(gamma), hip.ptr(beta),
                 hip.ptr(s1), hip.ptr(s2), inv, R, HEAD_HIDDEN, hip.ptr(dh16), st)
        _wgrad(dh16, x16, R, HEAD_HIDDEN, self.in_dim, self._w("0.weight", grad=True), ldx=3 * self.in_dim)
        dx = _rows_buf(R, self.in_dim, torch.float32, dev)
        _gemm(dh16, self._w("0.weight", transposed=True), R, self.in_dim, HEAD_HIDDEN, hip.EPI_F32, dx)
        self.saved = None
        return dx


_HEAD_PAD = os.environ.get("ATST_HEAD_PAD", "1") != "0"
_HEAD_SPLIT2 = os.environ.get("ATST_HEAD_SPLIT2", "gated")      # default of AtstEngine(head_split2=...): second head Linear in split-bf16: "gated" (only in front of another head), "all" (rounds 1-4), "none"
_PAD_FALLBACK_WARNED = False


def _gemm(A, B, M, N, K, epi, out):
    """out[M, N] = A[M, K] B[N, K]^T on the head path.  A and out are _rows_buf() views: above 4096 rows their allocations hold the row count rounded up
    to 4096, so the GEMM may run over the next multiple of 256 rows -- the geometry the 256 x 256 phased kernel takes (ATST-Frame's ~83 k masked rows are
    never such a multiple).  The extra output rows are computed from whatever the operand's padding rows hold and land in the output's padding rows; every
    consumer (BatchNorm sums, the next kernels) works on [:M]."""
    global _PAD_FALLBACK_WARNED
    Mp = -(-M // 256) * 256 if (_HEAD_PAD and M > 4096) else M
    if Mp != M:                                              # the padding rows must exist in both allocations (host-side check, no device work)
        # rows M .. Mp - 1 of A are uninitialised memory and their products land in out's padding rows: only an epilogue whose output rows are
        # independent of each other (plain fp32 store: no column sums, no amax, no row statistics) may run over them
        assert epi == hip.EPI_F32, "padded head GEMMs are fp32-store only"
        for t in (A, out):
            room = t.untyped_storage().nbytes() - t.storage_offset() * t.element_size()
            if room < Mp * t.stride(0) * t.element_size():
                Mp = M
                if not _PAD_FALLBACK_WARNED:
                    _PAD_FALLBACK_WARNED = True
                    warnings.warn("head GEMM operand is not a _rows_buf() view: running over %d rows on the 128-row kernel (slower)" % M)
    hip.call("atst_gemm_nt_bf16", hip.ptr(A), hip.ptr(B), Mp, N, K, K, K, epi, hip.ptr(out), N, None, None, None, None, 1,
             None, None, None, None, None, hip.stream())


def _wgrad(dY, X, M, N, K, dW, ldx=None):
    hip.call("atst_gemm_tn_bf16", hip.ptr(dY), hip.ptr(X), M, N, K, N, K if ldx is None else ldx, hip.ptr(dW), K, 0, hip.stream())


def as_one_buffer(views: Sequence[torch.Tensor]) -> Optional[torch.Tensor]:
    """If the views are consecutive contiguous slices of one device buffer (LogMelFrontend(out=...) wrote them there), the
    [sum(B_i), ...] tensor over that memory -- no torch.cat; otherwise None."""
    v0 = views[0]
    if not all(v.is_cuda and v.is_contiguous() and v.dtype == v0.dtype and v.shape[1:] == v0.shape[1:] for v in views):
        return None
    st, ptr = v0.untyped_storage().data_ptr(), v0.data_ptr()
    for v in views:
        if v.untyped_storage().data_ptr() != st or v.data_ptr() != ptr:
            return None
        ptr += v.numel() * v.element_size()
    return torch.as_strided(v0, (sum(v.shape[0] for v in views),) + tuple(v0.shape[1:]), v0.stride())


def group_views(widths: Sequence[int]) -> List[Tuple[int, int]]:
    """consecutive equal-width views share one encoder pass. ref: audiossl/models/atst/byol.py:107-112."""
    groups, start = [], 0
    for i in range(1, len(widths) + 1):
        if i == len(widths) or widths[i] != widths[start]:
            groups.append((start, i))
            start = i
    return groups


class AtstEngine:
    """Owns the flat parameter / gradient / optimizer-state buffers of student and teacher and runs the training step."""

    def __init__(self, arch: str = "small", frame: bool = False, depth: Optional[int] = None, ncrops: int = 2,
                 device: Optional[torch.device] = None, drop_path_rate: float = 0.1, n_pos: int = 251, fp8: bool = False,
                 symmetric: bool = True, patch_embed: str = "Linear", precise: bool = False, patch_h: int = 64, patch_w: int = 4,
                 head_split2: Optional[str] = None):
        if arch not in ARCH:
            raise RuntimeError("arch {} is not implemented".format(arch))      # ref: models/atst/atst.py:17
        hip.load()                                                               # fail loudly when the .so is missing
        if not torch.cuda.is_available():
            raise hip.HipError("AtstEngine needs a HIP device (MI355X); there is no CPU product path")
        self.arch, self.frame, self.ncrops = arch, frame, ncrops
        # Second Linear of a head on split-bf16 operands ([hi | lo | hi] x [hi | hi | lo], ~2^-16) or plain bf16: "gated" (default, round 5) = split only
        # where another head's BatchNorm + ReLU gates read the output (student projector); "all" = every head (rounds 1-4: parity runs); "none".
        # Measured deviation of "gated" from "all" on the goldens: DESIGN.md section 4 "Round 6".  Environment default: ATST_HEAD_SPLIT2.
        self.head_split2 = _HEAD_SPLIT2 if head_split2 is None else head_split2
        if self.head_split2 not in ("gated", "all", "none"):
            raise ValueError("head_split2 must be 'gated', 'all' or 'none'")
        if not symmetric and not frame:
            raise hip.HipError("symmetric=False is the ATST-Frame option (methods/atstframe/model.py:68-76)")
        self.symmetric = bool(symmetric)
        if patch_embed not in ("Linear", "CNN") or (patch_embed == "CNN" and not frame):
            raise NotImplementedError("patch_embed={} not implemted".format(patch_embed))      # ref: atstframe/audio_transformer.py:119-120
        self.patch_embed = patch_embed
        self.cfg = ARCH[arch]
        self.depth = self.cfg["depth"] if depth is None else depth
        self.n_pos = n_pos
        # patch geometry: ONE patch row of patch_h (= n_mels) bands x patch_w frames (the reference's --patch_h / --patch_w with
        # spec_h = n_mels, methods/atstframe/train.py:15,50-51; shipped recipes: 64 x 4; BASELINE configs[4]: 128 x 8 on 32 kHz audio)
        if (patch_h, patch_w) not in ((64, 4), (64, 8), (128, 4), (128, 8)):
            raise hip.HipError("supported patch geometries: 64 or 128 mel bands x 4 or 8 frames")
        self.patch_h, self.patch_w = patch_h, patch_w
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        self.layout = L = FlatLayout(arch, self.depth, frame, patch_embed, n_pos, patch_h, patch_w)
        dev = self.device
        z = lambda n, dt=torch.float32: torch.zeros(n, dtype=dt, device=dev)
        self.p32, self.g32, self.m32, self.v32 = z(L.n_student), z(L.n_student), z(L.n_student), z(L.n_student)
        self.t32 = z(L.n_teacher)
        self.p16, self.p16t, self.t16 = z(L.n_student, torch.bfloat16), z(L.n_student, torch.bfloat16), z(L.n_teacher, torch.bfloat16)
        # fp8 forward (BASELINE.json configs[4], ATST-base recipe): e4m3 shadows of the four Linear weights of every block,
        # per-tensor scaled, refreshed with the bf16 shadows; the backward and everything saved for it stay bf16
        self.fp8 = bool(fp8)
        # parity mode: training passes run on the fp32 / split-bf16 twin of the encoder (csrc/engine_hp.hip), everything else
        # (view grouping, heads -- already split-bf16 --, loss, optimizer, EMA) is this same engine.  17x slower (bench.py --precise).
        self.precise = bool(precise)
        if self.precise and self.fp8:
            raise hip.HipError("precise=True is the fp32 parity mode; it excludes fp8")
        if self.fp8:
            rows = []
            for i in range(self.depth):
                for nm in ("attn.qkv.weight", "attn.proj.weight", "mlp.fc1.weight", "mlp.fc2.weight"):
                    off, shape = L.entries[f"encoder.blocks.{i}.{nm}"]
                    rows.append((off, math.prod(shape)))
            self._f8_table = torch.tensor(rows, dtype=torch.int32, device=dev).contiguous()
            self.p8, self.t8 = z(L.n_student, torch.uint8), z(L.n_teacher, torch.uint8)
            self.dq_s, self.dq_t, self._f8_amax = z(len(rows)), z(len(rows)), z(len(rows))
            # clipped-element counters of the fixed forward activation scales, [student, teacher] (atst_encoder_t.f8_sat): never reset by
            # the step, read on demand (fp8_saturation()) -- a host read-back, so not once per step
            self.f8_sat = z(2, torch.int32)
            # forward activation sites, [student | teacher][depth][4] (LN1 out, attention out, LN2 out, GELU out): running scale
            # 448 / (margin * max amax over the window), initialised to the constants of round 2 / 3 (8, 8, 8, 4); amax of the current step
            self.f8a_scale = torch.tensor([8.0, 8.0, 8.0, 4.0], device=dev).repeat(2, self.depth).contiguous()
            self.f8a_amax = z(2 * 4 * self.depth).view(2, 4 * self.depth)
            # what the kernels post into: a site is AMAX_SITE_STRIDE floats (16 slots 256 B apart -- same-address atomics serialise at L2, and a
            # launch posts one per wave); reduced to f8a_amax / g8_amax once per step
            self.f8a_amax_sites = z(2 * 4 * self.depth * hip.AMAX_SITE_STRIDE).view(2, 4 * self.depth, hip.AMAX_SITE_STRIDE)
            self.f8a_hist = None                                     # [FP8_HISTORY, 2, 4 depth], allocated with the dgrad window below
            # fp8 dgrad (d = 768): e4m3 copy of the transposed weight shadows + delayed-scaling state of the four gradient operands
            # of every block ([depth][4]: g -> fc2, du -> fc1, g2 -> proj, dqkv -> qkv).  fp8_bwd_state: 0 off, 1 recording, 2 on.
            self.p8t = z(L.n_student, torch.uint8)
            self.g8_scale, self.g8_amax = torch.ones(4 * self.depth, device=dev), z(4 * self.depth)
            self.g8_amax_sites = z(4 * self.depth * hip.AMAX_SITE_STRIDE).view(4 * self.depth, hip.AMAX_SITE_STRIDE)
            self.fp8_bwd_state = 1 if (self.cfg["embed_dim"] in (384, 768) and os.environ.get("ATST_FP8_BWD", "1") != "0") else 0   # d = 384: round 6 (gemm_tn8 takes N, K % 128; the fp8 backward runs the unfused LayerNorm backward)
            # e4m3 weight gradients of fc1 / fc2 / proj (round 5; with the e4m3 dgrad only: they share its gradient-operand copies).  f8a_scale_used:
            # the student's forward activation scales as the forward of the current step used them (snapshot taken before they are advanced)
            self.fp8_wgrad = bool(self.fp8_bwd_state) and os.environ.get("ATST_FP8_WGRAD", "1") != "0"
            # ... and the qkv Linear: the NP = 256 attention backward writes dqkv as e4m3 only, the qkv dgrad and weight gradient read that copy
            # (gradient site 3).  fp8_qkv_state: 0 off, 1 recording site 3 (first step, or the first step after a checkpoint without it), 2 on.
            self.fp8_qkv_state = 1 if (self.fp8_wgrad and os.environ.get("ATST_FP8_QKV", "1") != "0") else 0
            # bf16 activation copies whose only readers would be bf16 weight gradients are not written once those run on the e4m3 copies
            self.fp8_lean = self.fp8_wgrad and os.environ.get("ATST_FP8_LEAN", "1") != "0"
            self.f8a_scale_used = self.f8a_scale[0].clone()
            self.fp8_margin = 2.0
            # amax HISTORY: the scale of a site is 448 / (margin * max amax over the last FP8_HISTORY steps), so one quiet step does not
            # shrink the headroom of the next (round-3 ADVICE: a single previous step with margin 2 saturates on any 2x spike); the
            # window, its cursor and the scales are optimizer state (checkpointed, broadcast on resume), and amax is MAX-reduced
            # over the ranks so that every replica quantises on the same grid.
            self.FP8_HISTORY = 16
            self.g8_hist, self._g8_hist_k = z(self.FP8_HISTORY * 4 * self.depth).view(self.FP8_HISTORY, 4 * self.depth), 0
            self.f8a_hist, self._f8a_hist_k = z(self.FP8_HISTORY * 8 * self.depth).view(self.FP8_HISTORY, 2, 4 * self.depth), 0
        self.bn_buffers: Dict[str, Dict[str, torch.Tensor]] = {}
        for key in ("student.projector", "student.predictor", "teacher.projector"):
            self.bn_buffers[key] = dict(running_mean=z(HEAD_HIDDEN), running_var=torch.ones(HEAD_HIDDEN, device=dev),
                                        num_batches_tracked=torch.zeros((), dtype=torch.int64, device=dev))
        # drop-path rates: torch.linspace(0, rate, depth) evaluated in fp32 like the reference (audio_transformer.py:107)
        self.dpr = [float(v) for v in torch.linspace(0, drop_path_rate, self.depth)]
        self._dpr_rates = torch.linspace(0, drop_path_rate, self.depth).to(self.device).view(-1, 1, 1)
        self.upload = Uploader(self.device)
        self.enc_off = self._build_offsets()
        flags = torch.zeros(L.n_student // ALIGN, dtype=torch.uint8)
        for name, (off, shape) in L.entries.items():
            n = math.prod(shape)
            decay = not (name.endswith(".bias") or len(shape) == 1)                  # ref: utils/common.py:58
            update = not (name == "encoder.mask_embed" and not frame)                 # grad is None in clip ATST
            ema = not name.startswith("predictor.")
            f = (1 if decay else 0) | (2 if update else 0) | (4 if ema else 0)
            flags[off // ALIGN:(off + n + ALIGN - 1) // ALIGN] = f
        self.flags = flags.to(dev)
        self.flags_ema_only = (flags & 4).to(dev)
        self.opt_step = 0
        self._passes: Dict[Tuple, EncoderPass] = {}
        self._cls_rows: Dict[Tuple[int, int], torch.Tensor] = {}
        self._infer_passes: "OrderedDict[Tuple, EncoderPass]" = OrderedDict()
        self._synced_version = (-1, -1)
        self.heads = {"teacher.projector": HeadPass(self, "teacher", "projector", self.cfg["embed_dim"]),
                      "student.projector": HeadPass(self, "student", "projector", self.cfg["embed_dim"]),
                      "student.predictor": HeadPass(self, "student", "predictor", HEAD_OUT)}
        self._acc = torch.zeros(1, device=dev)
        self._stats = torch.zeros(4, HEAD_OUT, device=dev)
        self._student_groups = None
        self._grads_summed = False
        self.overlap_teacher = os.environ.get("ATST_OVERLAP_T", "0") == "1"     # side-stream teacher pass next to the WHOLE student pass: measured no gain (full-chip kernels serialise), off by default
        # Two independent chains next to each other wherever one of them cannot fill the chip: the student's local-view groups (M = 26624: 208
        # blocks on 256 CUs, one 4-wave block each) run their forward beside the TEACHER pass (side stream) and their backward beside the
        # global-view group's backward.  Same box, same call: 4905-4921 -> 5142-5163 clips/s (+4.9 %).  ATST_OVERLAP_LT=0 switches it off (A/B).
        self.overlap_local_teacher = os.environ.get("ATST_OVERLAP_LT", "1") != "0"
        self._side = torch.cuda.Stream(device=self.device)
        self._comm = torch.cuda.Stream(device=self.device)     # gradient all-reduce underneath the backward pass
        self.overlap_comm = True
        self.grad_buckets = 4            # encoder slices of the overlapped all-reduce (+ one bucket for the two heads)

    # ---------------------------------------------------------------------------------------------------------------
    def _build_offsets(self) -> hip.EncOff:
        E, o = self.layout.entries, hip.EncOff()
        g = lambda n: E["encoder." + n][0]
        o.mask_embed, o.pos_embed = g("mask_embed"), g("pos_embed")
        o.cls_token = g("cls_token") if not self.frame else 0
        pe = "patch_embed.proj." if self.patch_embed == "CNN" else "patch_embed.patch_embed."
        o.patch_w, o.patch_b = g(pe + "weight"), g(pe + "bias")
        nf = "norm_frame" if self.frame else "norm"
        o.norm_w, o.norm_b = g(nf + ".weight"), g(nf + ".bias")
        for i in range(self.depth):
            b, l = f"blocks.{i}.", o.layer[i]
            l.ln1_w, l.ln1_b, l.qkv_w = g(b + "norm1.weight"), g(b + "norm1.bias"), g(b + "attn.qkv.weight")
            l.proj_w, l.proj_b = g(b + "attn.proj.weight"), g(b + "attn.proj.bias")
            l.ln2_w, l.ln2_b = g(b + "norm2.weight"), g(b + "norm2.bias")
            l.fc1_w, l.fc1_b = g(b + "mlp.fc1.weight"), g(b + "mlp.fc1.bias")
            l.fc2_w, l.fc2_b = g(b + "mlp.fc2.weight"), g(b + "mlp.fc2.bias")
        return o

    def param_view(self, net: str, name: str, grad: bool = False) -> torch.Tensor:
        off, shape = self.layout.entries[name]
        buf = (self.g32 if grad else self.p32) if net == "student" else self.t32
        return buf[off:off + math.prod(shape)].view(*shape)

    def load_weights(self, W: Dict[str, torch.Tensor]):
        """state_dict-keyed ('student.encoder...', 'teacher.projector...') fp32 tensors -> flat buffers + BN buffers."""
        with torch.no_grad():
            for net in ("student", "teacher"):
                for name in self.layout.entries:
                    if net == "teacher" and name.startswith("predictor."):
                        continue
                    self.param_view(net, name).copy_(W[f"{net}.{name}"].to(self.device))
            for key, bufs in self.bn_buffers.items():
                for b in bufs:
                    k = f"{key}.1.{b}"
                    if k in W:
                        bufs[b].copy_(W[k].to(self.device))
        self.sync_shadows(force=True)            # no collective here: replicas are aligned by ONE explicit broadcast_parameters()

    def init_weights(self, seed: int = 0):
        """Random init with the reference's distributions, drawn on the device: encoder Linear weights, cls/pos/mask
        tokens ~ trunc_normal(std 0.02, +-2 absolute), biases 0, LayerNorm (1,0)  (audio_transformer.py:116-129,
        modules/transformer.py:80-84); projector / predictor Linear keep torch's default kaiming-uniform(a=sqrt 5)
        = U(-1/sqrt(fan_in), 1/sqrt(fan_in)), BatchNorm (1,0)  (byol.py:6-22).  Teacher = copy of the student minus the
        predictor, BN buffers included (atst.py:22)."""
        g = torch.Generator(device=self.device).manual_seed(seed)
        with torch.no_grad():
            for name, (off, shape) in self.layout.entries.items():
                v = self.param_view("student", name)
                if name.startswith("encoder.patch_embed.proj."):        # nn.Conv2d keeps torch's default init (_init_weights skips it)
                    v.uniform_(-1.0 / 16.0, 1.0 / 16.0, generator=g)      # kaiming_uniform(a=sqrt 5): bound 1/sqrt(fan_in = 256), bias alike
                elif name.startswith("encoder."):
                    if name.endswith(".bias"):
                        v.zero_()
                    elif len(shape) == 1:
                        v.fill_(1.0)
                    else:
                        v.normal_(0.0, 0.02, generator=g).clamp_(-2.0, 2.0)
                else:
                    if name.endswith("1.weight"):
                        v.fill_(1.0)
                    elif name.endswith("1.bias"):
                        v.zero_()
                    else:
                        bound = 1.0 / math.sqrt(shape[1])
                        v.uniform_(-bound, bound, generator=g)
            self.t32.copy_(self.p32[:self.layout.n_teacher])
            for b in ("running_mean", "running_var", "num_batches_tracked"):
                self.bn_buffers["teacher.projector"][b].copy_(self.bn_buffers["student.projector"][b])
        self.sync_shadows(force=True)            # no collective here (a rank-0-only model must be constructible): see broadcast_parameters()

    def sync_shadows(self, force: bool = False):
        """Refresh bf16 shadows (and W^T copies) if the fp32 masters were modified through torch (version counters)."""
        ver = (self.p32._version, self.t32._version)
        if not force and ver == self._synced_version:
            return
        st = hip.stream()
        hip.call("atst_cast_bf16", hip.ptr(self.p32), self.p32.numel(), hip.ptr(self.p16), st)
        hip.call("atst_cast_bf16", hip.ptr(self.t32), self.t32.numel(), hip.ptr(self.t16), st)
        self._refresh_transposes()
        self._synced_version = (self.p32._version, self.t32._version)

    def _refresh_fp8(self):
        if not self.fp8:
            return
        n = self._f8_table.shape[0]
        for p32, p8, dq in ((self.p32, self.p8, self.dq_s), (self.t32, self.t8, self.dq_t)):
            hip.call("atst_quant_weights_fp8", hip.ptr(p32), hip.ptr(self._f8_table), n, hip.ptr(p8), hip.ptr(dq), hip.ptr(self._f8_amax),
                     hip.stream())

    def _refresh_fp8_transposed(self):
        """e4m3 copy of the transposed bf16 shadows (dgrad B operands), same per-tensor factors as the forward copies."""
        if self.fp8 and getattr(self, "fp8_bwd_state", 0):
            hip.call("atst_quant_bf16_table_fp8", hip.ptr(self.p16t), hip.ptr(self._f8_table), self._f8_table.shape[0], hip.ptr(self.dq_s),
                     hip.ptr(self.p8t), hip.stream())

    def _refresh_transposes(self):
        self._refresh_fp8()
        if getattr(self, "_tr_table", None) is None:
            rows, tiles = [], 0
            for name, (off, shape) in self.layout.entries.items():
                if len(shape) == 2:
                    rows.append((off, shape[0], shape[1], tiles))
                    tiles += -(-shape[0] // 64) * -(-shape[1] // 64)
            self._tr_table = torch.tensor(rows, dtype=torch.int32, device=self.device).contiguous()
            self._tr_tiles = tiles
        hip.call("atst_transpose_bf16_batch", hip.ptr(self.p16), hip.ptr(self.p16t), hip.ptr(self._tr_table),
                 self._tr_table.shape[0], self._tr_tiles, hip.stream())
        self._refresh_fp8_transposed()

    # ---------------------------------------------------------------------------------------------------------------
    def _pass(self, net: str, S: int, width: int, train: bool, slot: int) -> EncoderPass:
        key = (net, S, width, train, slot)
        if key not in self._passes:
            self._passes[key] = EncoderPass(self, net, S, width, train, precise=self.precise, record_amax=True)
        return self._passes[key]

    def inference_pass(self, net: str, S: int, width: int, keep: int = 4) -> EncoderPass:
        """Forward-only EncoderPass for the inference API (no activation tape: one layer of buffers re-used by every
        block, ~1/10 of a training workspace), least-recently-used eviction beyond `keep` geometries."""
        key = (net, S, width)
        lru = self._infer_passes
        if key in lru:
            lru.move_to_end(key)
        else:
            while len(lru) >= keep:
                lru.popitem(last=False)
            lru[key] = EncoderPass(self, net, S, width, False)
        return lru[key]

    def drop_path_scales(self, S: int, keep: Optional[torch.Tensor] = None) -> torch.Tensor:
        """[depth,2,S] fp32 factors keep/keep_prob.  keep (0/1, [depth,2,S]) may be injected for parity tests; otherwise
        drawn like the reference: floor(keep_prob + U[0,1)) per sample, per branch (modules/transformer.py:48-56)."""
        rates = self._dpr_rates
        if keep is None:                                            # drawn here: floor(1 - 0 + U[0, 1)) = 1 at rate 0 by itself
            keep = torch.floor((1.0 - rates) + torch.rand(self.depth, 2, S, device=self.device))
            return (keep / (1.0 - rates)).contiguous()
        keep = keep.to(self.device, torch.float32)
        # injected decisions: a block whose rate is 0 (block 0 of the linspace schedule) never drops in the reference, whatever was
        # drawn -- drop_path() returns its input when drop_prob == 0 (modules/transformer.py:49-50)
        return torch.where(rates > 0, keep / (1.0 - rates), torch.ones_like(keep)).contiguous()

    def _valid(self, lengths: torch.Tensor, use_cls: int, n_max: int = 1 << 30) -> torch.Tensor:
        """patch_length + CLS per sequence, clamped to the tokens the pass holds.  ref: audio_transformer.py:70-71,194."""
        l = torch.as_tensor(lengths)
        pw = self.patch_w
        if not l.is_cuda:                                           # host lengths (DataLoader batches): numpy -- see _frame_rows on why not CPU torch ops
            ln = l.numpy().astype(np.int64)
            return torch.from_numpy((np.minimum((ln - ln % pw) // pw, n_max - use_cls) + use_cls).astype(np.int32))
        l = l.to(torch.int64)
        v = torch.clamp((l - l % pw) // pw, max=n_max - use_cls) + use_cls
        return v.to(torch.int32).contiguous()                       # stays where `lengths` lives

    def _frame_rows(self, mk: torch.Tensor, valid: torch.Tensor, NP: int, mask_input: bool):
        """ATST-Frame row bookkeeping for one width group.  mk [S, n_tok] bool, valid [S] (frames per sequence); NP = the pass's ROW
        STRIDE between sequences (EncoderPass.RS: the padded token count, or the token count itself when short sequences are packed --
        every kernel indexes rows and rowflag by s * RS + n).
        Returns (rows int32 [R] device: flat token indices s*NP+n of masked valid frames in (b, n) order,
                 rowflag uint8 [S*NP] device or None).  With host-side masks (what the DataLoader delivers: the transform
        draws them with numpy, methods/atstframe/transform.py:84-101) everything is computed on the host and uploaded
        through pinned memory -- no device->host read-back, the row count R is known without touching the GPU."""
        S, n_tok = mk.shape
        if not mk.is_cuda and not valid.is_cuda:
            # numpy, not CPU torch ops: every CPU torch op on these 128 k-element tensors goes through the intra-op thread pool, and on a host shared with other
            # jobs each of them stalled for milliseconds (tools/debug/frame_host_profile2.py: 50 ms of host time per ATST-Frame step on a busy box against 3.8 ms on a
            # quiet one -- the GPU step is 51 ms: this was ATST-Frame's box-to-box spread)
            mkn, vn = mk.numpy(), valid.numpy()
            idx = np.flatnonzero(mkn & (np.arange(n_tok)[None, :] < vn[:, None]))
            rows = ((idx // n_tok) * NP + idx % n_tok).astype(np.int32)
            rowflag = None
            if mask_input:
                rf = np.zeros((S, NP), dtype=np.uint8)
                rf[:, :n_tok] = mkn
                rowflag = self.upload(torch.from_numpy(rf.reshape(-1)))
            return self.upload(torch.from_numpy(rows)), rowflag
        mk, valid = mk.to(self.device), valid.to(self.device)       # device-side masks: one read-back for the row count
        rowflag = None
        if mask_input:
            rowflag = torch.zeros(S, NP, dtype=torch.uint8, device=self.device)
            rowflag[:, :n_tok] = mk
            rowflag = rowflag.reshape(-1).contiguous()
        sel = mk & (torch.arange(n_tok, device=self.device)[None, :] < valid[:, None])
        s_idx, n_idx = sel.nonzero(as_tuple=True)                                         # row-major (b, n) order
        return (s_idx * NP + n_idx).to(torch.int32).contiguous(), rowflag

    def _run_net(self, net: str, mels, lengths, masks, mask_input: bool, keep, train: bool, after_group=None):
        """MultiCropWrapper.forward: encoder per width-group -> rows for the head (view-major).
        ref: audiossl/models/atst/byol.py:103-121 ; methods/atstframe/byol.py:118-138."""
        feats, groups = [], []
        use_cls = 0 if self.frame else 1
        for gi, (a, b) in enumerate(group_views([m.shape[-1] for m in mels])):
            grp = [m.to(self.device, torch.float32) for m in mels[a:b]]
            mel = grp[0].contiguous() if b - a == 1 else as_one_buffer(grp)
            if mel is None:
                mel = torch.cat(grp).contiguous()
            S, width = mel.shape[0], mel.shape[-1]
            ep = self._pass(net, S, width, train, gi)
            valid = self._valid(_host_cat([torch.as_tensor(l).reshape(-1) for l in lengths[a:b]]), use_cls, ep.n_tok + use_cls)
            rowflag = None
            if self.frame:
                mk = _host_cat([torch.as_tensor(m) for m in masks[a:b]])                      # [S, n_tok]
                if mk.dtype != torch.bool:
                    mk = mk.bool() if mk.is_cuda else torch.from_numpy(mk.numpy().astype(bool))
                rows, rowflag = self._frame_rows(mk, valid, ep.RS, mask_input)
            else:
                key = (S, ep.RS)
                if key not in self._cls_rows:
                    self._cls_rows[key] = (torch.arange(S, device=self.device, dtype=torch.int32) * ep.RS).contiguous()
                rows = self._cls_rows[key]
            dp = self.drop_path_scales(S, None if keep is None else keep[gi]) if self.dpr[-1] > 0 or keep is not None else None
            out16 = ep.forward(mel, valid, rowflag, dp)
            if ep.precise:                                  # fp32 rows: plain indexing (parity mode, speed is not the point)
                f = out16.index_select(0, rows.long())
            else:
                f = _rows_buf(rows.numel(), self.cfg["embed_dim"], torch.float32, self.device)
                hip.call("atst_gather_rows_bf16", hip.ptr(out16), hip.ptr(rows), rows.numel(), self.cfg["embed_dim"], hip.ptr(f), hip.stream())
            feats.append(f)
            groups.append((ep, rows))
            if after_group is not None:
                after_group(gi)
        return torch.cat(feats) if len(feats) > 1 else feats[0], groups

    def forward(self, mels: List[torch.Tensor], lengths: List[torch.Tensor], masks: Optional[List[torch.Tensor]] = None,
                keep_teacher=None, keep_student=None, train: bool = True):
        """teacher(first 2 views / unmasked) -> student(all views / masked) -> loss.  Returns (loss, std_s, std_t) as
        0-dim device tensors; saves what backward() needs.  ref: models/atst/atst.py:24-28, atstframe/model.py:68-72."""
        self.sync_shadows()
        mels = [m.to(self.device, torch.float32) for m in mels]
        asym = self.frame and not self.symmetric
        if asym:                                   # ref: atstframe/model.py:73-76 -- teacher: view 0 unmasked ; student: views 1.. masked
            t_sl, s_sl = slice(0, 1), slice(1, None)
        else:
            t_sl, s_sl = slice(0, len(mels) if self.frame else 2), slice(0, None)
        sub = lambda seq, sl: None if seq is None else list(seq)[sl]
        # The teacher pass has no data dependence on the student pass: it can run on a second HIP stream (measured: no gain,
        # full-chip kernels serialise; off by default).
        main = torch.cuda.current_stream()
        lt = self.overlap_local_teacher and not self.overlap_teacher and not self.frame and len(mels) > 2
        if self.overlap_teacher:
            self._side.wait_stream(main)
            with torch.cuda.stream(self._side):
                tf, _ = self._run_net("teacher", mels[t_sl], lengths[t_sl], sub(masks, t_sl), False, keep_teacher, False)
                t_out = self.heads["teacher.projector"].forward(tf, False)
        elif lt:
            t_out = None
            box = {}

            def teacher_next_to_local(gi):                     # called after the student's group gi has been enqueued on the main stream
                if gi == 0:
                    self._side.wait_stream(main)               # the teacher starts when the student's global-view group is done ...
                    with torch.cuda.stream(self._side):        # ... and runs while the main stream works through the local-view groups
                        box["tf"], _ = self._run_net("teacher", mels[t_sl], lengths[t_sl], sub(masks, t_sl), False, keep_teacher, False)
            sf, groups = self._run_net("student", mels[s_sl], lengths[s_sl], sub(masks, s_sl), True, keep_student, train, after_group=teacher_next_to_local)
            main.wait_stream(self._side)
            tf = box["tf"]
            tf.record_stream(main)
        else:
            tf, _ = self._run_net("teacher", mels[t_sl], lengths[t_sl], sub(masks, t_sl), False, keep_teacher, False)
            t_out = None
        if not lt:
            sf, groups = self._run_net("student", mels[s_sl], lengths[s_sl], sub(masks, s_sl), True, keep_student, train)
        if t_out is None:
            # the two projectors' BatchNorm statistics do not depend on each other: ONE cross-rank exchange for both
            # (2 forward SyncBN collectives per step instead of 3; the predictor's needs the student projector's output)
            tp, sp = self.heads["teacher.projector"], self.heads["student.projector"]
            st_t, st_s = parallel.combine_bn_stats_multi([tp.forward_stats(tf), sp.forward_stats(sf)])
            t_out = tp.forward_finish(*st_t, False)
            z = sp.forward_finish(*st_s, train)
        else:
            z = self.heads["student.projector"].forward(sf, train)
        s_out = self.heads["student.predictor"].forward(z, train)
        if self.overlap_teacher:
            main.wait_stream(self._side)
        self._teacher_keep = (tf, t_out)
        self._ds = _rows_buf(s_out.shape[0], s_out.shape[1], s_out.dtype, s_out.device)
        if asym:
            if s_out.shape[0] != t_out.shape[0]:
                raise hip.HipError("asymmetric ATST-Frame loss: teacher view 0 and student view 1 must select the same rows "
                                   "(one shared mask, equal lengths: methods/atstframe/transform.py:93-99)")
            B, ncrops, npairs = s_out.shape[0], -1, 1
        else:
            ncrops = 2 if self.frame else self.ncrops
            if s_out.shape[0] % ncrops or t_out.shape[0] % 2 or s_out.shape[0] // ncrops != t_out.shape[0] // 2:
                raise hip.HipError("views must contribute equal row counts (chunk() semantics of ByolLoss)")
            B, npairs = t_out.shape[0] // 2, 2 * ncrops - 2
        hip.call("atst_byol_loss_f32", hip.ptr(s_out), hip.ptr(t_out), B, ncrops, HEAD_OUT, hip.ptr(self._acc), hip.ptr(self._ds),
                 hip.ptr(self._stats), hip.stream())
        loss = 2.0 - 2.0 * self._acc[0] / (npairs * B)
        # one fused all-reduce instead of the reference's six (byol.py:48-50, twice)
        stats, ns, ntc = parallel.allreduce_monitor_sums(self._stats, float(s_out.shape[0]), float(t_out.shape[0]))
        self._student_groups = groups
        self.last_outputs = (s_out, t_out)
        self._fp8_after_forward()
        return loss, parallel.feature_std(stats[0], stats[1], ns), parallel.feature_std(stats[2], stats[3], ntc)

    def backward(self, grad_scale=1.0, zero_grad: bool = True):
        """Autograd of forward() wrt every student parameter, accumulated into the flat gradient buffer.
        grad_scale may be a python float or a 0-dim device tensor (the upstream d(loss))."""
        for ep, _ in (self._student_groups or ()):                # fail before anything is consumed (heads' saved state, gradient buffer)
            ep._check_lean()
        if zero_grad:
            self.g32.zero_()
        self._grads_summed = False
        if isinstance(grad_scale, torch.Tensor):
            ds = self._ds * grad_scale.to(self._ds.dtype)
        else:
            ds = self._ds if grad_scale == 1.0 else self._ds * float(grad_scale)
        dz = self.heads["student.predictor"].backward(ds)
        df = self.heads["student.projector"].backward(dz)
        groups, offs, r0 = list(self._student_groups), [], 0
        for ep, rows in groups:
            offs.append(r0)
            r0 += rows.numel()
        # Bucketed gradient all-reduce underneath the backward (DDP reducer semantics, methods/atst/train.py:19):
        #   bucket 0 = projector + predictor (final right here, before any encoder backward -- its reduction hides under
        #   the whole encoder backward); the view groups are then differentiated smallest first, and while the LAST
        #   (largest: the 10 s views) group walks down the blocks, the encoder is reduced in `grad_buckets` slices of
        #   blocks in reverse layer order, each as soon as that group has passed it (all other groups already have).
        order = sorted(range(len(groups)), key=lambda i: groups[i][0].M)
        self._async_reduce = False
        overlap = self.overlap_comm and parallel._collective() and zero_grad and not self.precise
        L = self.layout
        if overlap:
            self._reduce_async(L.entries["projector.0.weight"][0], L.n_student)
            self._async_reduce = True
        # The small (local-view) groups' backward on the side stream NEXT TO the large group's: their launches leave CUs idle, and every
        # gradient accumulation is an fp32 atomic (weight gradients, bias / LayerNorm column sums, token stage), so the two chains can run
        # concurrently.  The side stream is joined before the first encoder bucket is reduced (or at the end).
        main = torch.cuda.current_stream()
        side_bwd = self.overlap_local_teacher and len(order) > 1 and not self.precise and not self.frame
        joined = not side_bwd
        keep_alive = []
        side_events = {}
        for k, gi in enumerate(order):
            ep, rows = groups[gi]
            n = rows.numel()
            last = k == len(order) - 1
            src = df[offs[gi]:offs[gi] + n].contiguous()           # named: must outlive the launch call
            if side_bwd and not last:
                if k == 0:
                    self._side.wait_stream(main)                   # df is final on the main stream
                keep_alive.append(src)
                with torch.cuda.stream(self._side):
                    ep.dout.zero_()
                    hip.call("atst_scatter_rows_bf16", hip.ptr(src), hip.ptr(rows), n, self.cfg["embed_dim"], hip.ptr(ep.dout), hip.stream())
                    if overlap and k == len(order) - 2:
                        # the last side group walks the blocks in the SAME slices as the gradient buckets and leaves an event behind each: the
                        # communication stream then waits for "the side chain has passed slice j", not for the whole side chain (round 5;
                        # before, the first encoder bucket was handed over only after all 12 blocks of the side chain)
                        cuts = self.bucket_cuts()
                        for hi, lo in zip(cuts[:-1], cuts[1:]):
                            if hi == lo:
                                continue
                            ep.backward_range(lo, hi)
                            ev = torch.cuda.Event()
                            ev.record(self._side)
                            side_events[(hi, lo)] = ev
                    else:
                        ep.backward()
                continue
            ep.dout.zero_()
            if ep.precise:
                ep.dout.index_copy_(0, rows.long(), src)
            else:
                hip.call("atst_scatter_rows_bf16", hip.ptr(src), hip.ptr(rows), n, self.cfg["embed_dim"],
                         hip.ptr(ep.dout), hip.stream())
            if overlap and last:
                cuts = self.bucket_cuts()                                               # depth ... 0
                top = L.entries["projector.0.weight"][0]
                for hi, lo in zip(cuts[:-1], cuts[1:]):
                    if hi == lo:
                        continue
                    ep.backward_range(lo, hi)
                    a = 0 if lo == 0 else L.entries[f"encoder.blocks.{lo}.norm1.weight"][0]
                    # LN1 backward of block lo adds the fc2 bias gradient of block lo-1 (below the cut: reduced later)
                    ev = side_events.get((hi, lo))
                    if ev is None and not joined:
                        main.wait_stream(self._side)               # no per-slice events: every group has passed this slice only once the side chain is done
                        joined = True
                    self._reduce_async(a, top, ev)                 # (ev: the side chain's pass over this slice, awaited by the communication stream only)
                    top = a
            else:
                ep.backward()
        if not joined:
            main.wait_stream(self._side)
        for t in keep_alive:
            t.record_stream(self._side)
        self._fp8_after_backward()

    def bucket_cuts(self) -> List[int]:
        """Block indices at which the encoder gradient is cut into all-reduce buckets, descending from depth to 0.  grad_buckets
        even slices, then the LAST one (the only one whose reduction is not hidden under backward work) is cut again so that the
        exposed tail is block 0 + the token stage only (~7 % of the gradient for 12 blocks; it was blocks 0-2 = 25 %)."""
        nb = max(1, min(self.grad_buckets, self.depth))
        cuts = [round(self.depth * j / nb) for j in range(nb, -1, -1)]
        if len(cuts) >= 2 and cuts[-2] > 1:
            cuts.insert(-1, 1)
        return cuts

    def fp8_wgrad_mode(self) -> int:
        """atst_encoder_t.fp8_wgrad: 0 = bf16 weight gradients, 1 = e4m3 fc1 / fc2 / proj, 2 = + the e4m3 qkv gradient path, 3 = 1 + record site 3."""
        if not (self.fp8 and getattr(self, "fp8_wgrad", False)):
            return 0
        return {0: 1, 1: 3, 2: 2}[int(getattr(self, "fp8_qkv_state", 0))]

    def _fp8_after_backward(self):
        """Delayed scaling: this step's amax of every gradient operand becomes the next step's quantisation scale (448 / (margin amax));
        the first backward only records (bf16 dgrad), every later one runs the dgrad GEMMs on e4m3 operands."""
        if self.fp8 and getattr(self, "fp8_bwd_state", 0):
            torch.amax(self.g8_amax_sites, dim=-1, out=self.g8_amax)
            self.g8_amax_sites.zero_()
            if parallel._collective():
                dist.all_reduce(self.g8_amax, op=dist.ReduceOp.MAX)            # 4 * depth floats: every rank derives the same scales
            self.g8_hist[self._g8_hist_k % self.FP8_HISTORY].copy_(self.g8_amax)
            self._g8_hist_k += 1
            win = self.g8_hist.max(dim=0).values.contiguous()                   # sites never observed stay 0: their scale is left alone
            hip.call("atst_fp8_update_scales", hip.ptr(win), hip.ptr(self.g8_scale), win.numel(), float(self.fp8_margin), hip.stream())
            self.g8_amax.zero_()
            self.fp8_bwd_state = 2
            if getattr(self, "fp8_qkv_state", 0) == 1:
                self.fp8_qkv_state = 2

    def _fp8_after_forward(self):
        """Delayed scaling of the e4m3 forward: the amax every activation site recorded in this step's passes (student groups share
        their sites, the teacher has its own) enters a 16-step window; the next step quantises with 448 / (margin * window max).  Sites
        that saw nothing keep their scale.  MAX-reduced over the ranks (replicas quantise on one grid)."""
        if not self.fp8 or self.f8a_hist is None:
            return
        torch.amax(self.f8a_amax_sites, dim=-1, out=self.f8a_amax)
        self.f8a_amax_sites.zero_()
        if parallel._collective():
            dist.all_reduce(self.f8a_amax, op=dist.ReduceOp.MAX)
        self.f8a_hist[self._f8a_hist_k % self.FP8_HISTORY].copy_(self.f8a_amax)
        self._f8a_hist_k += 1
        win = self.f8a_hist.max(dim=0).values.contiguous()
        hip.call("atst_fp8_update_scales", hip.ptr(win), hip.ptr(self.f8a_scale), win.numel(), float(self.fp8_margin), hip.stream())
        self.f8a_amax.zero_()

    def fp8_saturation(self, reset: bool = False) -> Dict[str, int]:
        """Activation elements clipped at +-448 by the e4m3 forward since the last reset, per network.  The forward scales are constants
        (csrc/engine.hip ACT_SCALE: LayerNorm / attention outputs beyond +-56, GELU outputs beyond +-112 saturate); a non-zero count
        says the run has outgrown them.  Synchronises (one 8-byte read-back): call it at logging cadence, not per step."""
        if not self.fp8:
            return {"student": 0, "teacher": 0}
        v = self.f8_sat.tolist()
        if reset:
            self.f8_sat.zero_()
        return {"student": int(v[0]), "teacher": int(v[1])}

    def fp8_state(self) -> Optional[dict]:
        """Delayed-scaling state of the fp8 path (optimizer state: saved with the moments, see trainer.save_checkpoint).  The forward activation
        scales, their 16-step window and its cursor belong to EVERY fp8 engine (ATST-small fp8, or ATST_FP8_BWD=0, quantise their forward too: a
        resumed run must quantise as the saved one did -- ADVICE r4); the g8_* keys of the e4m3 dgrad only exist when that path is on."""
        if not self.fp8:
            return None
        st = {"f8a_scale": self.f8a_scale.cpu(), "f8a_hist": self.f8a_hist.cpu(), "f8a_hist_k": int(self._f8a_hist_k)}
        if getattr(self, "fp8_bwd_state", 0):
            st.update({"g8_scale": self.g8_scale.cpu(), "g8_hist": self.g8_hist.cpu(), "hist_k": int(self._g8_hist_k), "state": int(self.fp8_bwd_state),
                       "qkv_state": int(getattr(self, "fp8_qkv_state", 0))})
        return st

    def load_fp8_state(self, st: Optional[dict]):
        if not st or not self.fp8:
            return
        if "f8a_scale" in st:
            self.f8a_scale.copy_(st["f8a_scale"]); self.f8a_hist.copy_(st["f8a_hist"]); self._f8a_hist_k = int(st["f8a_hist_k"])
        if "g8_scale" in st and getattr(self, "fp8_bwd_state", 0):
            self.g8_scale.copy_(st["g8_scale"]); self.g8_hist.copy_(st["g8_hist"])
            self._g8_hist_k, self.fp8_bwd_state = int(st["hist_k"]), int(st["state"])
            if getattr(self, "fp8_qkv_state", 0):                        # a checkpoint from before the qkv path has no site-3 scale: record one step first
                self.fp8_qkv_state = 2 if int(st.get("qkv_state", 0)) == 2 else 1

    def _reduce_async(self, a: int, b: int, also=None):
        """Sum g32[a:b] over ranks on the communication stream, ordered after everything enqueued so far on the current stream (and after
        the event `also`, recorded on another stream that accumulates into the same slice)."""
        main = torch.cuda.current_stream()
        self._comm.wait_stream(main)
        if also is not None:
            self._comm.wait_event(also)
        with torch.cuda.stream(self._comm):
            parallel.allreduce_sum_(self.g32[a:b])

    def broadcast_parameters(self, optimizer_state: bool = False):
        """DDP init: every rank takes rank 0's parameters / BN buffers (and, after a resume, the optimizer moments and the
        AdamW step count), so the replicas start identical whatever their seeds or checkpoints were.  A COLLECTIVE: every
        rank must call it, exactly once, from the same place (Trainer.fit start, bench.py, the multi-rank tests) -- it is
        deliberately not hidden inside init_weights() / load_weights().  ref: DDP wrapper of Trainer(strategy="ddp"),
        methods/atst/train.py:18-32.  No-op without a process group."""
        if not parallel._collective():
            return
        bufs = [self.p32, self.t32] + [t for b in self.bn_buffers.values() for t in b.values()]
        f8a = bool(optimizer_state and self.fp8)                                          # forward scales: every fp8 engine
        f8 = bool(f8a and getattr(self, "fp8_bwd_state", 0))                              # e4m3 dgrad state: only when that path is on
        if optimizer_state:
            bufs += [self.m32, self.v32]
        if f8a:
            bufs += [self.f8a_scale, self.f8a_hist]
        if f8:
            bufs += [self.g8_scale, self.g8_hist]
        for t in bufs:
            dist.broadcast(t, 0)
        if optimizer_state:
            step = torch.tensor([self.opt_step, self._g8_hist_k if f8 else 0, self.fp8_bwd_state if f8 else 0, self._f8a_hist_k if f8a else 0,
                                 getattr(self, "fp8_qkv_state", 0) if f8 else 0], dtype=torch.int64, device=self.device)
            dist.broadcast(step, 0)
            self.opt_step = int(step[0].item())
            if f8a:
                self._f8a_hist_k = int(step[3].item())
            if f8:
                self._g8_hist_k, self.fp8_bwd_state = int(step[1].item()), int(step[2].item())
                if getattr(self, "fp8_qkv_state", 0):
                    self.fp8_qkv_state = 2 if int(step[4].item()) == 2 else 1
        self.sync_shadows(force=True)

    def allreduce_grads(self):
        """DDP: sum student gradients over ranks (RCCL over xGMI); the 1/world mean is folded into the optimizer kernel.
        When backward() already started the bucketed reduction, this only joins the communication stream."""
        if getattr(self, "_async_reduce", False):
            torch.cuda.current_stream().wait_stream(self._comm)
            self._async_reduce = False
            self._grads_summed = True
            return
        self._grads_summed = parallel.allreduce_sum_(self.g32)

    def optimizer_step(self, lr: float, wd: float, ema_m: Optional[float], betas=(0.9, 0.999), eps: float = 1e-6):
        """HF-AdamW over the two parameter groups + EMA teacher + bf16 shadow refresh in one pass.
        ref: transformers AdamW via methods/atst/model.py:44-48 ; atst.py:29-34."""
        if getattr(self, "_async_reduce", False):              # bucketed reduction still in flight: join it first
            self.allreduce_grads()
        self.opt_step += 1
        t = self.opt_step
        step_size = lr * math.sqrt(1.0 - betas[1] ** t) / (1.0 - betas[0] ** t)
        do_ema = ema_m is not None
        hip.call("atst_adamw_ema_step", hip.ptr(self.p32), hip.ptr(self.g32), hip.ptr(self.m32), hip.ptr(self.v32),
                 hip.ptr(self.t32) if do_ema else None, hip.ptr(self.p16), hip.ptr(self.t16) if do_ema else None,
                 hip.ptr(self.flags), self.layout.n_student, self.layout.n_teacher, lr, wd, betas[0], betas[1], eps, step_size,
                 ema_m if do_ema else 1.0, (1.0 / self.world) if self._grads_summed else 1.0, hip.stream())
        self._refresh_transposes()
        self._synced_version = (self.p32._version, self.t32._version)

    def ema_update(self, m: float):
        """ATST.update_teacher(m) alone (compat path when an external optimizer stepped the student)."""
        self.sync_shadows()
        hip.call("atst_adamw_ema_step", hip.ptr(self.p32), hip.ptr(self.g32), hip.ptr(self.m32), hip.ptr(self.v32), hip.ptr(self.t32),
                 hip.ptr(self.p16), hip.ptr(self.t16), hip.ptr(self.flags_ema_only), self.layout.n_student, self.layout.n_teacher,
                 0.0, 0.0, 0.9, 0.999, 1e-6, 0.0, m, 1.0, hip.stream())
        self._refresh_fp8()
