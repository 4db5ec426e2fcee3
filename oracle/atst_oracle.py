"""CPU oracle for the ATST / ATST-Frame pre-training step.

TEST INFRASTRUCTURE ONLY.  This file is a plain-PyTorch fp32 *restatement* (functional style, no
nn.Module, no Lightning) of the algorithm on the north_star hot path.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it; the product path in
``audiossl_amd/`` never does and fails loudly when the HIP library is missing.

Pinning status (see DESIGN.md section "Oracle"):
  * encoder / head / loss / EMA / schedules / param groups / mask application: pinned against goldens
    produced by importing the reference in the build container (tests/golden/make_golden.py).
  * mel front end (torchaudio), block-mask sampler (fairseq), HF-AdamW (transformers<5): the arithmetic
    lives in third-party packages that are absent from /root/reference and from this image ->
    "parity unpinned" for those three; they follow the published algorithms and are
    pinned only by known-answer tests (tests/test_oracle_mel.py) and by cross-checks against independent third-party
    implementations that ARE in the image: transformers.audio_utils (mel), transformers' wav2vec2 _compute_mask_indices
    (a port of the fairseq sampler), torch.optim.AdamW (up to HF-AdamW's two documented differences) --
    tests/test_oracle_mel.py, tests/test_host_logic_cpu.py.

All ``ref:`` citations are relative to the upstream repository root.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor
Weights = Dict[str, Tensor]

# ----------------------------------------------------------------------------------------------------------------------
# constants of the path
# ----------------------------------------------------------------------------------------------------------------------
SAMPLE_RATE = 16000
N_FFT = 1024
HOP = 160
N_MELS = 64
F_MIN, F_MAX = 60.0, 7800.0
TOP_DB = 80.0
DB_MIN, DB_MAX = -79.6482, 50.6842          # ref: audiossl/methods/atst/transform.py:18
PATCH_H, PATCH_W = 64, 4                    # ref: audiossl/models/atst/audio_transformer.py:367-371
MASK_NEG = -10000.0                         # ref: audiossl/modules/transformer.py:157
LN_EPS = 1e-6                               # ref: audio_transformer.py:368 (partial(nn.LayerNorm, eps=1e-6))
BN_EPS, BN_MOMENTUM = 1e-5, 0.1             # torch BatchNorm1d defaults, ref: byol.py:15

ARCH = {  # ref: audio_transformer.py:367-374
    "small": dict(embed_dim=384, depth=12, num_heads=6),
    "base": dict(embed_dim=768, depth=12, num_heads=12),
}


# ----------------------------------------------------------------------------------------------------------------------
# bf16 emulation mode (test infrastructure for tests/test_step_gpu.py / tools/bf16_floor.py).
# With ``emulate_bf16()`` active the SAME fp32 restatement rounds to bfloat16 (round-to-nearest-even) exactly where the HIP
# path does -- GEMM operands (weights' bf16 shadows, LayerNorm outputs, qkv, softmax probabilities, attention output, GELU
# output), the saved pre-activation u that GELU' is evaluated on, and every gradient tensor that the HIP backward hands
# to a GEMM as a bf16 operand (d(LN out), dqkv, dS, d(attention out), du, the DropPath-scaled branch gradients, the head
# gradients) -- while everything the HIP path keeps in fp32 stays fp32 (residual stream, accumulators, softmax, LayerNorm /
# BatchNorm statistics, the split-bf16 head Linears' forward).  It separates "bf16 rounding" from "error" in the
# HIP-vs-reference gradient differences: HIP vs this mode must agree to ~1e-3, this mode vs plain fp32 is the bf16 floor.
# ----------------------------------------------------------------------------------------------------------------------
_EMU = False
_EMU_OFF: frozenset = frozenset()
# Rounding sites of the encoder (names for emulate_bf16(off=...), tools/parity_report.py --by-site).  Forward values: "w" (bf16 weight shadows), "patches"
# (patch-embed operand), "ln1" / "ln2" / "ln_final" (LayerNorm outputs = GEMM operands), "qkv" (qkv GEMM output), "P" (softmax probabilities), "attn_out"
# (attention output = proj operand), "gelu_out" (a = fc2 operand), "u_saved" (the bf16 copy of the pre-activation GELU' is evaluated on).  Gradient operands:
# "g_ln1" / "g_ln2" / "g_final" (d LayerNorm output), "g_qkv", "g_S" (d scores), "g_attn_out", "g_proj_out" / "g_fc2_out" (branch gradients = dY of the
# proj / fc2 backward), "g_fc1_out" (du), "g_patch".
ROUNDING_SITES = ("w", "patches", "ln1", "qkv", "P", "attn_out", "ln2", "gelu_out", "u_saved", "ln_final",
                  "g_ln1", "g_qkv", "g_S", "g_attn_out", "g_proj_out", "g_ln2", "g_fc1_out", "g_fc2_out", "g_patch", "g_final")


class emulate_bf16:
    """off: rounding sites (ROUNDING_SITES) left in fp32 -- attribution of the bf16 gradient error to the places that round."""

    def __init__(self, on: bool = True, off=()):
        self.on, self.off = on, frozenset(off)
        assert self.off <= set(ROUNDING_SITES), self.off - set(ROUNDING_SITES)

    def __enter__(self):
        global _EMU, _EMU_OFF
        self.prev, _EMU = (_EMU, _EMU_OFF), self.on
        _EMU_OFF = self.off
        return self

    def __exit__(self, *a):
        global _EMU, _EMU_OFF
        _EMU, _EMU_OFF = self.prev


_GATES: Dict[str, Tensor] = {}


class relu_gates:
    """Evaluate the heads with GIVEN ReLU gate patterns ({'student.projector.': bool [R, 4096], ...}) instead of the sign of
    the oracle's own BatchNorm output.  The gates are the only discontinuity of the training step: a 1e-3 forward
    perturbation flips ~0.1 % of them and moves every upstream gradient by several per cent.  With the gates of the HIP run
    injected, the remaining HIP-vs-oracle gradient difference is the smooth part (what tolerance tests can bound)."""

    def __init__(self, gates: Dict[str, Tensor]):
        self.gates = gates

    def __enter__(self):
        global _GATES
        self.prev, _GATES = _GATES, dict(self.gates)
        return self

    def __exit__(self, *a):
        global _GATES
        _GATES = self.prev


def _bf(x: Tensor) -> Tensor:
    return x.to(torch.bfloat16).to(torch.float32)


class _RoundFwd(torch.autograd.Function):           # value rounded, gradient passed through (a bf16 operand / saved tensor)
    @staticmethod
    def forward(ctx, x):
        return _bf(x)

    @staticmethod
    def backward(ctx, g):
        return g


class _RoundGrad(torch.autograd.Function):          # value untouched, gradient rounded (a bf16 gradient operand)
    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return _bf(g)


class _GeluSavedBf16(torch.autograd.Function):      # a = gelu(u) on the fp32 accumulator; GELU' on the bf16 copy of u that is saved
    @staticmethod
    def forward(ctx, u):
        ctx.save_for_backward(u if "u_saved" in _EMU_OFF else _bf(u))
        return F.gelu(u)

    @staticmethod
    def backward(ctx, g):
        (u,) = ctx.saved_tensors
        cdf = 0.5 * (1.0 + torch.erf(u * 0.7071067811865476))
        pdf = torch.exp(-0.5 * u * u) * 0.3989422804014327
        return g * (cdf + u * pdf)


class _HeadLinear(torch.autograd.Function):         # forward ~exact (split-bf16 MFMA, 2^-16); backward on bf16 operands
    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(_bf(x), _bf(w))
        return x @ w.t()

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        g = _bf(g)
        return g @ w, g.t() @ x


_FP8 = False
_FP8_ACT = {}
_FP8_RESID16 = False                                 # emulate_fp8(resid_bf16=True): no-grad passes keep their residual stream in bf16 (csrc/engine.hip, round 6)
FP8_ACT_SCALE, FP8_ACT_SCALE_GELU = 8.0, 4.0        # csrc/engine.hip ACT_SCALE / ACT_SCALE_GELU


class emulate_fp8:
    """On top of emulate_bf16: the four Linear layers of every block evaluate their FORWARD on OCP e4m3 copies of their
    operands -- activations x fixed scale, weights x 448 / amax per tensor, both saturated at +-448 -- exactly like the HIP
    fp8 path (csrc/engine.hip gemm8, csrc/optim.hip quant kernels); gradients flow as if the bf16 operands had been used."""

    def __init__(self, on: bool = True, act_scales=None, resid_bf16: bool = True):
        """act_scales: optional {weight key: scale of that Linear's INPUT activation} -- the running (delayed) scales of a later step
        (AtstEngine.f8a_scale); sites not listed use the constants FP8_ACT_SCALE / FP8_ACT_SCALE_GELU (= the first step).
        resid_bf16 (round 6): passes evaluated WITHOUT gradient (teacher / inference) keep their residual stream in bf16 -- the sum of every residual
        add is rounded -- like the HIP fp8 inference passes (csrc/engine.hip `xb16`; tuning hook 2100 switches it off there, False here)."""
        self.on, self.act_scales, self.resid_bf16 = on, dict(act_scales or {}), resid_bf16

    def __enter__(self):
        global _FP8, _FP8_ACT, _FP8_RESID16
        self.prev, _FP8 = (_FP8, _FP8_ACT, _FP8_RESID16), self.on
        _FP8_ACT, _FP8_RESID16 = self.act_scales, self.resid_bf16
        return self

    def __exit__(self, *a):
        global _FP8, _FP8_ACT, _FP8_RESID16
        _FP8, _FP8_ACT, _FP8_RESID16 = self.prev


def _q8(x: Tensor, scale) -> Tensor:
    return (x * scale).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).to(torch.float32) / scale


_FP8_BWD = None                                     # emulate_fp8_dgrad instance while active


class emulate_fp8_dgrad:
    """On top of emulate_bf16 + emulate_fp8: the dgrad GEMMs of fc2, fc1 and proj (d = 768: csrc/engine.hip encoder_bwd_range, `use8`)
    run on OCP e4m3 copies of their operands.  The gradient operand of site k (= the bf16 gradient arriving at that Linear's output) is
    quantised with a DELAYED scale -- 448 / (margin * amax of the same site in the previous step), csrc/optim.hip fp8_update_scales --
    and the transposed weight shadow with the per-tensor factor of the forward copy (448 / amax of the fp32 master, applied to the bf16
    shadow: atst_quant_bf16_table_fp8).  Every weight gradient stays on bf16 operands unless wgrad=True (round 5: csrc/gemm_tn8.hip): then the
    weight gradients of the e4m3 sites are products of the SAME e4m3 gradient operand and of the e4m3 activation copy the forward used (x
    quantised with its forward activation scale).  The qkv Linear stays on bf16 operands unless qkv=True (round 5: the NP = 256 attention
    backward then writes dqkv as e4m3 only, csrc/attention.hip attn_bwd256_kernel<2>): it becomes a fourth site with the same rules.
      scales = None  -> record only (what the engine's first backward does): bf16 dgrad, self.amax[site] filled
      scales = {site: s} -> e4m3 dgrad at those sites with those scales; self.amax again holds this step's amax.
    `site` is the weight key ("student.encoder.blocks.0.mlp.fc2.weight").  next_scales() turns the recorded amax into the next step's
    scales exactly like the device kernel."""
    SITES = ("mlp.fc2.weight", "mlp.fc1.weight", "attn.proj.weight")

    def __init__(self, scales=None, margin: float = 2.0, wgrad: bool = False, qkv: bool = False):
        self.scales, self.margin, self.amax, self.wgrad = scales, margin, {}, wgrad
        self.sites = self.SITES + (("attn.qkv.weight",) if qkv else ())

    def next_scales(self):
        return {k: 448.0 / (self.margin * v) for k, v in self.amax.items() if v > 0}

    def __enter__(self):
        global _FP8_BWD
        self.prev, _FP8_BWD = _FP8_BWD, self
        return self

    def __exit__(self, *a):
        global _FP8_BWD
        _FP8_BWD = self.prev


class _Fp8Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, w_master, act_scale, key):
        ctx.save_for_backward(x, w, w_master)
        ctx.key, ctx.act_scale = key, act_scale
        ws = 448.0 / w_master.abs().max().clamp_min(1e-30)          # the e4m3 shadow is cut from the fp32 master
        return _q8(x, act_scale) @ _q8(w_master, ws).t()

    @staticmethod
    def backward(ctx, g):
        x, w, w_master = ctx.saved_tensors                           # weight gradient: as if the bf16 operands had been used
        dw = g.reshape(-1, g.shape[-1]).t() @ x.reshape(-1, x.shape[-1])
        st, key = _FP8_BWD, ctx.key
        if st is not None and key.endswith(st.sites):
            st.amax[key] = max(st.amax.get(key, 0.0), float(g.abs().max()))      # g is the bf16 gradient operand (emulate_bf16 rounds it)
            if st.scales is not None and key in st.scales:
                ws = 448.0 / w_master.abs().max().clamp_min(1e-30)
                g8 = _q8(g, st.scales[key])
                if st.wgrad:                                         # e4m3 weight gradient: the same gradient copy x the forward's activation copy
                    dw = g8.reshape(-1, g.shape[-1]).t() @ _q8(x, ctx.act_scale).reshape(-1, x.shape[-1])
                return g8 @ _q8(w, ws), dw, None, None, None         # w = the bf16 shadow (its transposed copy is what gets quantised)
        return g @ w, dw, None, None, None


def _linear(x: Tensor, W: Weights, key: str, b: Optional[Tensor], act_scale: float = FP8_ACT_SCALE) -> Tensor:
    """F.linear on the (bf16-shadow when emulating) weight W[key], or its e4m3-forward version inside emulate_fp8()."""
    if not _FP8:
        return F.linear(x, _w(W, key), b)
    y = _Fp8Linear.apply(x, _w(W, key), W[key].detach(), _FP8_ACT.get(key, act_scale), key)
    return y if b is None else y + b


def _r(x, site=None):
    return _RoundFwd.apply(x) if (_EMU and site not in _EMU_OFF) else x


def _rg(x, site=None):
    return _RoundGrad.apply(x) if (_EMU and x.requires_grad and site not in _EMU_OFF) else x


def _rb(x, site=None, gsite=None):
    return _r(x, site) if gsite == "-" else _rg(_r(x, site), gsite)      # "-": this gradient is not rounded on the HIP side


def _w(W: Weights, k: str) -> Tensor:
    """a GEMM weight operand: the bf16 shadow when emulating."""
    return _r(W[k], "w")


# ----------------------------------------------------------------------------------------------------------------------
# a1-a3: log-mel front end (torchaudio MelSpectrogram -> AmplitudeToDB -> MinMax).  PARITY UNPINNED (torchaudio absent)
# ----------------------------------------------------------------------------------------------------------------------
def hz_to_mel_htk(f):
    return 2595.0 * np.log10(1.0 + np.asarray(f, dtype=np.float64) / 700.0)


def mel_to_hz_htk(m):
    return 700.0 * (10.0 ** (np.asarray(m, dtype=np.float64) / 2595.0) - 1.0)


def mel_filterbank(n_freqs: int = N_FFT // 2 + 1, f_min: float = F_MIN, f_max: float = F_MAX,
                   n_mels: int = N_MELS, sample_rate: int = SAMPLE_RATE) -> Tensor:
    """HTK triangular filterbank, no area normalisation -> [n_freqs, n_mels] fp32.
    Follows torchaudio.functional.melscale_fbanks (call site ref: methods/atst/transform.py:14-15).
    torchaudio evaluates this in fp32 torch ops; it is evaluated the same way here."""
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m_min = 2595.0 * math.log10(1.0 + f_min / 700.0)
    m_max = 2595.0 * math.log10(1.0 + f_max / 700.0)
    m_pts = torch.linspace(m_min, m_max, n_mels + 2)
    f_pts = 700.0 * (10.0 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    down = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return torch.clamp(torch.min(down, up), min=0.0)


def hann_window_padded(win_length: int = N_FFT, n_fft: int = N_FFT) -> Tensor:
    """Periodic Hann of win_length, zero-padded centred to n_fft (what torch.stft does when win<n_fft)."""
    w = torch.hann_window(win_length, periodic=True)
    if win_length < n_fft:
        left = (n_fft - win_length) // 2
        w = F.pad(w, (left, n_fft - win_length - left))
    return w


def power_spectrogram(wave: Tensor, win_length: int = N_FFT) -> Tensor:
    """wave [..., L] -> |STFT|^2 [..., 513, 1 + L // 160]; center=True, reflect pad, onesided."""
    shape = wave.shape
    x = wave.reshape(-1, shape[-1])
    spec = torch.stft(x, n_fft=N_FFT, hop_length=HOP, win_length=win_length,
                      window=torch.hann_window(win_length, periodic=True), center=True, pad_mode="reflect",
                      normalized=False, onesided=True, return_complex=True)
    p = spec.real ** 2 + spec.imag ** 2
    return p.reshape(*shape[:-1], p.shape[-2], p.shape[-1])


def log_mel(wave: Tensor, win_length: int = N_FFT, n_mels: int = N_MELS, sample_rate: int = SAMPLE_RATE) -> Tensor:
    """wave [B, L] (one clip-view per row) -> normalised log-mel [B, 1, n_mels, T].
    a1: MelSpectrogram; a2: AmplitudeToDB(power, top_db=80) with ONE max per clip-view; a3: MinMax.
    sample_rate / n_mels: the reference's `sr` / `n_mels` constructor parameters (methods/atstframe/transform.py:14-16); the
    sample rate only changes the filterbank (hop, n_fft, f_min, f_max are fixed there)."""
    fb = mel_filterbank(n_mels=n_mels, sample_rate=sample_rate)
    power = power_spectrogram(wave.float(), win_length)                       # [B,513,T]
    mel = torch.matmul(power.transpose(-1, -2), fb).transpose(-1, -2)         # [B,64,T]
    db = 10.0 * torch.log10(torch.clamp(mel, min=1e-10))
    db = db - 10.0 * math.log10(max(1e-10, 1.0))
    floor = db.amax(dim=(-2, -1), keepdim=True) - TOP_DB
    db = torch.max(db, floor)
    out = (db - DB_MIN) / (DB_MAX - DB_MIN) * 2.0 - 1.0                      # ref: transforms/common.py:97-110
    return out.unsqueeze(1)


# ----------------------------------------------------------------------------------------------------------------------
# a4: patchify (integer, bit-exact) + patch_length
# ----------------------------------------------------------------------------------------------------------------------
def patchify(mel: Tensor, patch_h: int = PATCH_H, patch_w: int = PATCH_W) -> Tensor:
    """[S,1,H,W] -> [S, (W//pw)*(H//ph), ph*pw]; patch index = w*(H//ph)+h ; in-patch index k = f*pw + t.
    ref: audio_transformer.py:56-75 (einops 'b c (h p1) (w p2) -> b (w h) (p1 p2 c)')."""
    S, C, H, W = mel.shape
    assert C == 1
    Hh, Ww = H - H % patch_h, W - W % patch_w
    x = mel[:, 0, :Hh, :Ww].reshape(S, Hh // patch_h, patch_h, Ww // patch_w, patch_w)   # s h p1 w p2
    x = x.permute(0, 3, 1, 2, 4)                                                           # s w h p1 p2
    return x.reshape(S, (Ww // patch_w) * (Hh // patch_h), patch_h * patch_w)


def patch_length(length: Tensor, spec_h: int = 64, patch_h: int = PATCH_H, patch_w: int = PATCH_W) -> Tensor:
    """ref: audio_transformer.py:70-71."""
    hh = (spec_h - spec_h % patch_h) // patch_h
    return hh * ((length - length % patch_w) // patch_w)


# ----------------------------------------------------------------------------------------------------------------------
# a6-a11: encoder
# ----------------------------------------------------------------------------------------------------------------------
def drop_path_rates(depth: int, rate: float = 0.1) -> List[float]:
    """ref: audio_transformer.py:107 (torch.linspace(0, rate, depth), .item() -> python floats of fp32 values)."""
    return [float(v) for v in torch.linspace(0, rate, depth)]


def key_padding_bias(n_tokens: int, valid: Tensor) -> Tensor:
    """[S] int -> additive [S,1,1,N] fp32 ; -10000 where key index >= valid.  ref: transformer.py:152-159."""
    idx = torch.arange(n_tokens)
    return (idx[None, :] >= valid[:, None]).float()[:, None, None, :] * MASK_NEG


def block_forward(W: Weights, pre: str, x: Tensor, bias: Optional[Tensor], num_heads: int,
                  keep_attn: Optional[Tensor] = None, keep_mlp: Optional[Tensor] = None,
                  drop_prob: float = 0.0) -> Tensor:
    """Pre-LN block.  keep_* : optional [S] {0,1} DropPath keep decisions (injected, because GPU and CPU RNG
    streams differ); output scaled by 1/(1-drop_prob).  ref: transformer.py:95-150."""
    S, N, C = x.shape
    hd = C // num_heads
    # d = 384: the HIP dgrad GEMM in front of a LayerNorm runs that LayerNorm's backward in its epilogue, on the fp32 accumulators -- the gradient of the
    # LayerNorm output is never rounded to bf16 there (csrc/gemm.hip EPI_LNBWD); at d = 768 it passes through HBM as bf16 (sites g_ln1 / g_ln2).  The e4m3 recording
    # step at d = 384 (first backward) runs the separate pass and does round: one step, not modelled.
    g_ln1, g_ln2 = ("-", "-") if C == 384 else ("g_ln1", "g_ln2")
    h = _rb(F.layer_norm(x, (C,), W[pre + "norm1.weight"], W[pre + "norm1.bias"], LN_EPS), "ln1", g_ln1)
    qkv = _rb(_linear(h, W, pre + "attn.qkv.weight", W.get(pre + "attn.qkv.bias")), "qkv", "g_qkv")
    qkv = qkv.reshape(S, N, 3, num_heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    att = torch.matmul(q, k.transpose(-2, -1)) * (hd ** -0.5)
    if bias is not None:
        att = att + bias
    att = _r(_rg(att, "g_S").softmax(dim=-1), "P")
    y = _rb(torch.matmul(att, v).transpose(1, 2).reshape(S, N, C), "attn_out", "g_attn_out")
    y = _rg(_linear(y, W, pre + "attn.proj.weight", W[pre + "attn.proj.bias"]), "g_proj_out")
    if keep_attn is not None and drop_prob > 0.0:
        y = y / (1.0 - drop_prob) * keep_attn.to(y.dtype)[:, None, None]
    r16 = _FP8 and _FP8_RESID16 and not torch.is_grad_enabled()     # fp8 inference / teacher passes: bf16 residual stream (rounded at every add)
    x = _bf(x + y) if r16 else x + y
    h = _rb(F.layer_norm(x, (C,), W[pre + "norm2.weight"], W[pre + "norm2.bias"], LN_EPS), "ln2", g_ln2)
    h = _rg(_linear(h, W, pre + "mlp.fc1.weight", W[pre + "mlp.fc1.bias"]), "g_fc1_out")
    h = _r(_GeluSavedBf16.apply(h), "gelu_out") if _EMU else F.gelu(h)      # exact erf GELU (nn.GELU default)
    h = _rg(_linear(h, W, pre + "mlp.fc2.weight", W[pre + "mlp.fc2.bias"], FP8_ACT_SCALE_GELU), "g_fc2_out")
    if keep_mlp is not None and drop_prob > 0.0:
        h = h / (1.0 - drop_prob) * keep_mlp.to(h.dtype)[:, None, None]
    return _bf(x + h) if r16 else x + h


def encoder_tokens(W: Weights, pre: str, mel: Tensor, length: Optional[Tensor], use_cls: bool,
                   mask_index: Optional[Tensor] = None, mask_input: bool = True) -> Tuple[Tensor, Optional[Tensor]]:
    """patch-embed + (mask-token blend) + CLS + positional table ("cut").
    ref: audio_transformer.py:153-186 (clip) ; methods/atstframe/audio_transformer.py:161-181 (frame)."""
    # patch geometry from the shapes: one patch row of patch_h = spec_h bands (reference --patch_h with spec_h = n_mels) x patch_w frames
    ph = mel.shape[2]
    pw = W[pre + "patch_embed.patch_embed.weight"].shape[1] // ph
    patches = _r(patchify(mel, ph, pw), "patches")
    x = _rg(F.linear(patches, _w(W, pre + "patch_embed.patch_embed.weight"), W[pre + "patch_embed.patch_embed.bias"]), "g_patch")
    S, T, C = x.shape
    plen = patch_length(length, mel.shape[2], ph, pw) if length is not None else None
    if mask_index is not None and mask_input:
        m = mask_index.unsqueeze(2).expand(S, T, C).float()
        x = (1 - m) * x + m * W[pre + "mask_embed"].expand(S, T, C)
    if use_cls:
        x = torch.cat((W[pre + "cls_token"].expand(S, -1, -1), x), dim=1)
        x = x + W[pre + "pos_embed"][:, :T + 1, :]
    else:
        x = x + W[pre + "pos_embed"][:, 1:T + 1, :]
    return x, plen


def encoder_forward(W: Weights, pre: str, mel: Tensor, length: Tensor, arch: str = "small", depth: Optional[int] = None,
                    use_cls: bool = True, mask_index: Optional[Tensor] = None, mask_input: bool = True,
                    keep: Optional[Tensor] = None, drop_path_rate: float = 0.1,
                    return_blocks: bool = False):
    """Clip encoder (use_cls=True) returns CLS [S,C]  (ref: audio_transformer.py:188-221);
    frame encoder (use_cls=False) returns LN(x)[mask & valid] [M,C]  (ref: atstframe/audio_transformer.py:183-207).
    keep: optional [depth, 2, S] DropPath keep decisions (0/1)."""
    cfg = ARCH[arch]
    depth = cfg["depth"] if depth is None else depth
    x, plen = encoder_tokens(W, pre, mel, length, use_cls, mask_index, mask_input)
    valid = plen + 1 if use_cls else plen
    bias = key_padding_bias(x.shape[1], valid)
    dpr = drop_path_rates(depth, drop_path_rate)
    outs = []
    for i in range(depth):
        ka = keep[i, 0] if keep is not None else None
        km = keep[i, 1] if keep is not None else None
        x = block_forward(W, f"{pre}blocks.{i}.", x, bias, cfg["num_heads"], ka, km, dpr[i])
        if return_blocks:
            outs.append(x)
    C = x.shape[-1]
    if use_cls:
        y = _rb(F.layer_norm(x, (C,), W[pre + "norm.weight"], W[pre + "norm.bias"], LN_EPS), "ln_final", "g_final")
        out = y[:, 0]
    else:
        y = _rb(F.layer_norm(x, (C,), W[pre + "norm_frame.weight"], W[pre + "norm_frame.bias"], LN_EPS), "ln_final", "g_final")
        lm = torch.arange(x.shape[1])[None, :] < plen[:, None]
        out = y[mask_index & lm]
    return (out, outs) if return_blocks else out


# ----------------------------------------------------------------------------------------------------------------------
# a12-a14: head, multi-view wrapper, loss
# ----------------------------------------------------------------------------------------------------------------------
def frame_intermediate_layers(W: Weights, pre: str, mel: Tensor, length: Tensor, n: int = 1, scene: bool = True,
                              arch: str = "small", depth: Optional[int] = None) -> Tensor:
    """FrameAST.get_intermediate_layers in eval mode (ref: methods/atstframe/audio_transformer.py:259-281): norm_frame of
    the last n block outputs; scene=True -> masked mean over the valid frames ([S, n*C]), else the frames ([S, T, n*C])."""
    cfg = ARCH[arch]
    depth = cfg["depth"] if depth is None else depth
    x, plen = encoder_tokens(W, pre, mel, length, False, None, False)
    bias = key_padding_bias(x.shape[1], plen)
    C = x.shape[-1]
    outs = []
    for i in range(depth):
        x = block_forward(W, f"{pre}blocks.{i}.", x, bias, cfg["num_heads"], None, None, 0.0)
        if depth - i <= n:
            y = F.layer_norm(x, (C,), W[pre + "norm_frame.weight"], W[pre + "norm_frame.bias"], LN_EPS)
            if scene:
                lm = (torch.arange(x.shape[1])[None, :] < plen[:, None]).unsqueeze(-1)
                outs.append((y * lm).sum(1) / (plen.unsqueeze(-1) + 1e-6))
            else:
                outs.append(y)
    return torch.cat(outs, dim=-1)


def mlp_head(W: Weights, pre: str, x: Tensor, update_running: bool = True) -> Tensor:
    """Linear(no bias) -> BatchNorm1d(train mode, batch statistics) -> ReLU -> Linear(no bias).
    Updates running_mean / running_var / num_batches_tracked in W in place like nn.BatchNorm1d does.
    ref: models/atst/byol.py:6-22."""
    lin = (lambda a, w: _HeadLinear.apply(a, w)) if _EMU else F.linear
    h = lin(x, W[pre + "0.weight"])
    rm = W[pre + "1.running_mean"] if update_running else None
    rv = W[pre + "1.running_var"] if update_running else None
    h = F.batch_norm(h, rm, rv, W[pre + "1.weight"], W[pre + "1.bias"], True, BN_MOMENTUM, BN_EPS)
    if update_running and (pre + "1.num_batches_tracked") in W:
        W[pre + "1.num_batches_tracked"] += 1
    gate = _GATES.get(pre) if _GATES else None
    h = h * gate.to(h.dtype) if gate is not None else F.relu(h)     # injected ReLU gates: see relu_gates()
    return lin(h, W[pre + "3.weight"])


def group_views(widths: Sequence[int]) -> List[Tuple[int, int]]:
    """Consecutive equal-width views are concatenated on batch and share one encoder pass.
    ref: byol.py:107-112 (unique_consecutive + cumsum)."""
    groups, start = [], 0
    for i in range(1, len(widths) + 1):
        if i == len(widths) or widths[i] != widths[start]:
            groups.append((start, i))
            start = i
    return groups


def net_forward(W: Weights, pre: str, mels: List[Tensor], lengths: List[Tensor], arch: str, predictor: bool,
                keep: Optional[List[Tensor]] = None, depth: Optional[int] = None,
                drop_path_rate: float = 0.1, update_running: bool = True) -> Tensor:
    """MultiCropWrapper.forward (clip).  keep: one [depth,2,S_group] tensor per width-group.
    ref: byol.py:103-121."""
    feats = []
    for gi, (a, b) in enumerate(group_views([m.shape[-1] for m in mels])):
        feats.append(encoder_forward(W, pre + "encoder.", torch.cat(mels[a:b]), torch.cat(lengths[a:b]), arch,
                                     depth=depth, keep=None if keep is None else keep[gi],
                                     drop_path_rate=drop_path_rate))
    out = mlp_head(W, pre + "projector.", torch.cat(feats), update_running)
    if predictor:
        out = mlp_head(W, pre + "predictor.", out, update_running)
    return out


def feature_std(y: Tensor) -> Tensor:
    """Monitor: mean over dims of sqrt(unbiased var of L2-normalised rows + 1e-6) (world size 1).
    ref: byol.py:42-53, 62-63."""
    y = F.normalize(y, dim=-1)
    y = y.reshape(-1, y.shape[-1])
    n = y.shape[0]
    zs, zss = y.sum(0), (y ** 2).sum(0)
    var = zss / (n - 1) - zs ** 2 / (n * (n - 1))
    return torch.sqrt(var + 1e-6).mean()


def byol_pair_loss(p: Tensor, z: Tensor) -> Tensor:
    """ref: byol.py:24-41 (simplified=False)."""
    return 2 - 2 * (F.normalize(p, dim=-1) * F.normalize(z, dim=-1)).sum(dim=1).mean()


def byol_loss(student: Tensor, teacher: Tensor, ncrops: int) -> Tuple[Tensor, Tensor, Tensor]:
    """Cross-view loss: mean over (teacher view iq in 0..1, student view iv in 0..ncrops-1, iq != iv).
    ref: byol.py:57-78."""
    std_s, std_t = feature_std(student), feature_std(teacher)
    sv = student.chunk(ncrops)
    tv = teacher.detach().chunk(2)
    total, n = 0.0, 0
    for iq, q in enumerate(tv):
        for iv, v in enumerate(sv):
            if iq == iv:
                continue
            total = total + byol_pair_loss(q, v)
            n += 1
    return total / n, std_s, std_t


def atst_forward(W: Weights, mels: List[Tensor], lengths: List[Tensor], arch: str = "small", ncrops: int = 2,
                 keep_teacher=None, keep_student=None, depth: Optional[int] = None, drop_path_rate: float = 0.1):
    """ATST.forward: teacher on the first two views, student on all views.  W holds 'student.*' and 'teacher.*'.
    ref: models/atst/atst.py:24-28."""
    with torch.no_grad():
        t = net_forward(W, "teacher.", mels[:2], lengths[:2], arch, False, keep_teacher, depth, drop_path_rate)
    s = net_forward(W, "student.", mels, lengths, arch, True, keep_student, depth, drop_path_rate)
    return byol_loss(s, t, ncrops)


# -- ATST-Frame --------------------------------------------------------------------------------------------------------
def frame_net_forward(W: Weights, pre: str, mels, lengths, masks, mask_input: bool, arch: str, predictor: bool,
                      keep=None, depth=None, drop_path_rate: float = 0.1) -> Tensor:
    """ref: methods/atstframe/byol.py:118-138."""
    feats = []
    for gi, (a, b) in enumerate(group_views([m.shape[-1] for m in mels])):
        feats.append(encoder_forward(W, pre + "encoder.", torch.cat(mels[a:b]), torch.cat(lengths[a:b]), arch,
                                     depth=depth, use_cls=False, mask_index=torch.cat(masks[a:b]),
                                     mask_input=mask_input, keep=None if keep is None else keep[gi],
                                     drop_path_rate=drop_path_rate))
    out = mlp_head(W, pre + "projector.", torch.cat(feats))
    if predictor:
        out = mlp_head(W, pre + "predictor.", out)
    return out


def frame_byol_loss(student: Tensor, teacher: Tensor):
    """symmetric=True branch.  ref: methods/atstframe/byol.py:61-84."""
    std_s, std_t = feature_std(student), feature_std(teacher)
    sv, tv = student.chunk(2), teacher.chunk(2)
    total, n = 0.0, 0
    for iq, q in enumerate(tv):
        for iv, v in enumerate(sv):
            if iq == iv:
                continue
            total = total + byol_pair_loss(q, v)
            n += 1
    return total / n, std_s, std_t


def frame_atst_forward(W: Weights, mels, lengths, masks, arch: str = "small", keep_teacher=None, keep_student=None,
                       depth=None, drop_path_rate: float = 0.1, symmetric: bool = True):
    """FrameATST.forward: teacher sees the unmasked input, student the mask-token-substituted one.  symmetric: both views
    through both networks (model.py:68-72); otherwise teacher on view 0, student on view 1, one pair scored with
    2 - 2 cosine_similarity (model.py:73-76, byol.py:23-36,83-84)."""
    if symmetric:
        with torch.no_grad():
            t = frame_net_forward(W, "teacher.", mels, lengths, masks, False, arch, False, keep_teacher, depth, drop_path_rate)
        s = frame_net_forward(W, "student.", mels, lengths, masks, True, arch, True, keep_student, depth, drop_path_rate)
        return frame_byol_loss(s, t)
    with torch.no_grad():
        t = frame_net_forward(W, "teacher.", mels[:1], lengths[:1], masks[:1], False, arch, False, keep_teacher, depth, drop_path_rate)
    s = frame_net_forward(W, "student.", mels[1:], lengths[1:], masks[1:], True, arch, True, keep_student, depth, drop_path_rate)
    loss = 2 - 2 * F.cosine_similarity(s, t, dim=-1).mean()
    return loss, feature_std(s), feature_std(t)


# ----------------------------------------------------------------------------------------------------------------------
# a15-a16: EMA, schedules, parameter groups, HF-AdamW
# ----------------------------------------------------------------------------------------------------------------------
def teacher_param_names(W: Weights) -> List[str]:
    """Teacher tensors touched by the EMA: encoder.* and projector.* *parameters* (BN buffers are not).
    ref: atst.py:29-34."""
    return [k for k in W if k.startswith("teacher.") and not k.endswith(("running_mean", "running_var", "num_batches_tracked"))]


@torch.no_grad()
def ema_update(W: Weights, m: float) -> None:
    for k in teacher_param_names(W):
        W[k].mul_(m).add_((1 - m) * W["student." + k[len("teacher."):]])


def cosine_scheduler_step(base_value, final_value, max_steps, warmup_steps=0, start_warmup_value=0) -> np.ndarray:
    """ref: utils/common.py:29-39."""
    warm = np.linspace(start_warmup_value, base_value, warmup_steps) if warmup_steps > 0 else np.array([])
    it = np.arange(max_steps - warmup_steps)
    sched = final_value + 0.5 * (base_value - final_value) * (1 + np.cos(np.pi * it / len(it)))
    out = np.concatenate((warm, sched))
    assert len(out) == max_steps
    return out


def param_groups(named_shapes: Sequence[Tuple[str, Tuple[int, ...]]]) -> Tuple[List[str], List[str]]:
    """(regularised, not_regularised) names: biases and all 1-D tensors get no weight decay.
    ref: utils/common.py:41-68."""
    reg, noreg = [], []
    for name, shape in named_shapes:
        (noreg if (name.endswith(".bias") or len(shape) == 1) else reg).append(name)
    return reg, noreg


@torch.no_grad()
def hf_adamw_step(p: Tensor, g: Tensor, m: Tensor, v: Tensor, step: int, lr: float, wd: float,
                  beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-6) -> None:
    """transformers.optimization.AdamW (<5.0) single-tensor update; ``step`` is 1-based.  PARITY UNPINNED.
    call site ref: methods/atst/model.py:44-48 ; algorithm: SURVEY.md Appendix A.4."""
    m.mul_(beta1).add_(g, alpha=1.0 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1.0 - beta2)
    denom = v.sqrt().add_(eps)
    step_size = lr * math.sqrt(1.0 - beta2 ** step) / (1.0 - beta1 ** step)
    p.addcdiv_(m, denom, value=-step_size)
    if wd > 0.0:
        p.add_(p, alpha=-lr * wd)


# ----------------------------------------------------------------------------------------------------------------------
# a20: mask samplers.  block sampler PARITY UNPINNED (fairseq absent); random sampler restates random_mask.py:27-36
# ----------------------------------------------------------------------------------------------------------------------
def block_mask(num_patches: int = 250, mask_prob: float = 0.65, mask_length: int = 5, min_masks: int = 2,
               rng: Optional[np.random.RandomState] = None) -> np.ndarray:
    """fairseq compute_mask_indices(shape=(1,S), mask_type="static", no_overlap=False, min_space=0) for bsz 1.
    call site ref: methods/atstframe/random_mask.py:5-15 ; algorithm: SURVEY.md Appendix A.3."""
    rng = np.random if rng is None else rng
    S = num_patches
    num = int(mask_prob * S / float(mask_length) + rng.rand())
    num = max(min_masks, num)
    lengths = np.full(num, mask_length)
    min_len = int(min(lengths)) if num > 0 else 0
    if S - min_len <= num:
        min_len = S - num - 1
    starts = rng.choice(S - min_len, num, replace=False)
    idx = np.asarray([starts[j] + o for j in range(num) for o in range(lengths[j])])
    idx = np.unique(idx[idx < S])
    mask = np.zeros(S, dtype=bool)
    mask[idx] = True
    return mask


def random_mask(num_patches: int, available: int, ratio: float, generator: Optional[torch.Generator] = None) -> Tensor:
    """ref: random_mask.py:27-30 (get_mask_one)."""
    m = torch.randperm(available, generator=generator) < available * ratio
    return F.pad(m, (0, num_patches - available), value=True)


# ----------------------------------------------------------------------------------------------------------------------
# deterministic weight recipe shared by the golden generator, the tests, bench.py and smoke()
# ----------------------------------------------------------------------------------------------------------------------
def encoder_shapes(arch: str, depth: Optional[int] = None, frame: bool = False, n_pos: int = 251, patch_h: int = PATCH_H, patch_w: int = PATCH_W):
    cfg = ARCH[arch]
    d, depth = cfg["embed_dim"], (cfg["depth"] if depth is None else depth)
    shapes = [("mask_embed", (1, 1, d))]
    if not frame:
        shapes.append(("cls_token", (1, 1, d)))
    shapes += [("pos_embed", (1, n_pos, d)),
               ("patch_embed.patch_embed.weight", (d, patch_h * patch_w)), ("patch_embed.patch_embed.bias", (d,))]
    for i in range(depth):
        b = f"blocks.{i}."
        shapes += [(b + "norm1.weight", (d,)), (b + "norm1.bias", (d,)),
                   (b + "attn.qkv.weight", (3 * d, d)),
                   (b + "attn.proj.weight", (d, d)), (b + "attn.proj.bias", (d,)),
                   (b + "norm2.weight", (d,)), (b + "norm2.bias", (d,)),
                   (b + "mlp.fc1.weight", (4 * d, d)), (b + "mlp.fc1.bias", (4 * d,)),
                   (b + "mlp.fc2.weight", (d, 4 * d)), (b + "mlp.fc2.bias", (d,))]
    nf = "norm_frame" if frame else "norm"
    shapes += [(nf + ".weight", (d,)), (nf + ".bias", (d,))]
    return shapes


def head_shapes(in_dim: int, hidden: int = 4096, out: int = 256):
    return [("0.weight", (hidden, in_dim)), ("1.weight", (hidden,)), ("1.bias", (hidden,)),
            ("1.running_mean", (hidden,)), ("1.running_var", (hidden,)), ("1.num_batches_tracked", ()),
            ("3.weight", (out, hidden))]


def student_shapes(arch: str, depth: Optional[int] = None, frame: bool = False, patch_h: int = PATCH_H, patch_w: int = PATCH_W):
    """state_dict order of reference ``model.student`` (encoder, projector, predictor)."""
    d = ARCH[arch]["embed_dim"]
    out = [("encoder." + k, s) for k, s in encoder_shapes(arch, depth, frame, patch_h=patch_h, patch_w=patch_w)]
    out += [("projector." + k, s) for k, s in head_shapes(d)]
    out += [("predictor." + k, s) for k, s in head_shapes(256)]
    return out


def recipe_weights(arch: str = "small", depth: Optional[int] = None, frame: bool = False, seed: int = 0,
                   perturb: bool = True, patch_h: int = PATCH_H, patch_w: int = PATCH_W) -> Weights:
    """Deterministic numpy (PCG64) weights for student+teacher, state_dict-keyed ('student.*', 'teacher.*').
    NOT the reference's init distribution (that needs torch's RNG stream); it is a fixed, well-conditioned
    recipe that both the golden generator (loading it into the reference model) and the tests rebuild.
    With perturb=True LN/BN affine parameters and biases are non-trivial so that every code path is exercised."""
    rng = np.random.default_rng(seed)
    W: Weights = {}
    for name, shape in student_shapes(arch, depth, frame, patch_h, patch_w):
        if name.endswith("num_batches_tracked"):
            t = np.zeros((), dtype=np.int64)
        elif name.endswith("running_mean"):
            t = np.zeros(shape, dtype=np.float32)
        elif name.endswith("running_var"):
            t = np.ones(shape, dtype=np.float32)
        elif len(shape) == 1 and name.endswith(".weight"):           # LN / BN scale
            t = (1.0 + (0.1 * rng.standard_normal(shape) if perturb else 0.0)) * np.ones(shape)
        elif name.endswith(".bias"):
            t = 0.02 * rng.standard_normal(shape) if perturb else np.zeros(shape)
        elif name.startswith(("projector.", "predictor.")):          # kaiming-uniform-like scale
            bound = 1.0 / math.sqrt(shape[1])
            t = rng.uniform(-bound, bound, size=shape)
        else:
            t = 0.02 * rng.standard_normal(shape)
        W["student." + name] = torch.from_numpy(np.asarray(t, dtype=np.float32 if t.dtype != np.int64 else np.int64)).clone()
    for k in list(W):
        if "predictor" not in k:
            W["teacher." + k[len("student."):]] = W[k].clone()
    return W


def recipe_mel(n_seq: int, width: int = 1001, seed: int = 1, n_mels: int = 64) -> Tensor:
    """Synthetic normalised log-mel in about [-1,1] (smooth + noise), fp32 [n_seq,1,n_mels,width]."""
    rng = np.random.default_rng(seed)
    base = rng.standard_normal((n_seq, 1, n_mels, 1)) * 0.3 + rng.standard_normal((n_seq, 1, 1, width)) * 0.3
    x = base + 0.3 * rng.standard_normal((n_seq, 1, n_mels, width))
    return torch.from_numpy(np.clip(x, -1.0, 1.0).astype(np.float32))


def recipe_wave(n_clips: int, n_samples: int = 160000, seed: int = 1234) -> Tensor:
    """BASELINE.md section 3 / SURVEY.md 8(d) synthetic waveform: N(0, 0.1^2) clipped to [-1,1]."""
    g = torch.Generator().manual_seed(seed)
    return torch.clamp(0.1 * torch.randn(n_clips, n_samples, generator=g), -1.0, 1.0)


# ---- host augmentations -------------------------------------------------------------------------
def random_resize_crop(lms: Tensor, i: int, j: int, h: int, w: int, virtual_crop_scale=(1.0, 1.5)) -> Tensor:
    """RandomResizeCrop.forward with the (i, j, h, w) of get_params given (ref: transforms/byol_a.py:33-49). lms [C,H,W]."""
    C, H, W = lms.shape
    CH, CW = int(H * virtual_crop_scale[0]), int(W * virtual_crop_scale[1])
    canvas = torch.zeros(C, CH, CW, dtype=torch.float32)
    x0, y0 = (CW - W) // 2, (CH - H) // 2
    canvas[:, y0:y0 + H, x0:x0 + W] = lms
    crop = canvas[:, i:i + h, j:j + w]
    return F.interpolate(crop.unsqueeze(0), size=(H, W), mode="bicubic", align_corners=True).squeeze(0).float()


def log_mixup_exp(x: Tensor, z: Tensor, a: float, start: int = 0) -> Tensor:
    """Mixup.forward's mix of input x [C,H,Wx] with bank entry z [C,H,Wz] at ratio a = ratio * U (ref: byol_a.py:61-83 called
    as log_mixup_exp(x, z, 1 - a) at :104): log((1-a) e^x + a e^z + eps); the shorter one goes into the window
    [start, start + min(Wx, Wz)) of the longer one."""
    ex, ez = x.exp(), z.exp()
    Wx, Wz = x.shape[-1], z.shape[-1]
    eps = torch.finfo(torch.float32).eps
    if Wx < Wz:
        return torch.log((1.0 - a) * ex + a * ez[..., start:start + Wx] + eps)
    if Wx > Wz:
        ex = ex.clone()
        ex[..., start:start + Wz] = (1.0 - a) * ex[..., start:start + Wz] + a * ez
        return torch.log(ex + eps)
    return torch.log((1.0 - a) * ex + a * ez + eps)
