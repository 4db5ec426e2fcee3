"""ctypes binding of libatst_hip.so (C ABI in include/atst_hip.h).

torch is used for device memory and streams only; every tensor crosses the boundary as ``data_ptr()`` + sizes.
There is NO fallback: if the shared object is missing or a kernel reports an error, this raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libatst_hip.so")
if os.environ.get("ATST_LIB_TAG"):          # experiment builds side by side (audiossl_amd/build.py): A/B runs inside one gpurun call
    LIB_PATH = os.path.join(_HERE, "lib", f"libatst_hip_{os.environ['ATST_LIB_TAG']}.so")
ATST_MAX_DEPTH = 24
ABI_VERSION = 120                                     # include/atst_hip.h ATST_ABI_VERSION these bindings were written for

EPI_BF16, EPI_F32, EPI_BIAS_GELU, EPI_RESID, EPI_DGELU, EPI_PATCH, EPI_LNBWD = range(7)
AMAX_SLOTS, AMAX_SLOT_STRIDE = 16, 64                 # include/atst_hip.h ATST_AMAX_*: a running-amax site is 16 slots, 256 B apart
AMAX_SITE_STRIDE = AMAX_SLOTS * AMAX_SLOT_STRIDE


class HipError(RuntimeError):
    pass


class Wgrad8(C.Structure):
    """atst_wgrad8_t (include/atst_hip.h)"""
    _fields_ = [("dY8", C.c_void_p), ("X8", C.c_void_p), ("dW", C.c_void_p), ("N", C.c_int), ("K", C.c_int), ("ldy", C.c_int), ("ldx", C.c_int),
                ("ldw", C.c_int), ("scale_y", C.c_void_p), ("scale_x", C.c_void_p)]


class Wgrad(C.Structure):
    """atst_wgrad_t (include/atst_hip.h)."""
    _fields_ = [("dY", C.c_void_p), ("X", C.c_void_p), ("dW", C.c_void_p)] + [(n, C.c_int) for n in ("M", "N", "K", "ldy", "ldx", "ldw")]


class LayerOff(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("ln1_w", "ln1_b", "qkv_w", "proj_w", "proj_b", "ln2_w", "ln2_b", "fc1_w", "fc1_b",
                                         "fc2_w", "fc2_b")]


class EncOff(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("mask_embed", "cls_token", "pos_embed", "patch_w", "patch_b", "norm_w", "norm_b")] + \
               [("layer", LayerOff * ATST_MAX_DEPTH)]


class Encoder(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("S", "NP", "n_tok", "width", "C", "H", "depth", "use_cls", "train")] + \
               [("p32", C.c_void_p), ("p16", C.c_void_p), ("p16t", C.c_void_p), ("g32", C.c_void_p), ("off", EncOff),
                ("mel", C.c_void_p), ("valid", C.c_void_p), ("rowflag", C.c_void_p), ("dp_scale", C.c_void_p),
                ("ws", C.c_void_p), ("ws_bytes", C.c_size_t), ("tap", C.c_void_p), ("tap_first", C.c_int),
                ("p8", C.c_void_p), ("w_dq", C.c_void_p), ("fp8", C.c_int), ("patch_h", C.c_int), ("patch_w", C.c_int),
                ("p8t", C.c_void_p), ("g8_scale", C.c_void_p), ("g8_amax", C.c_void_p), ("fp8_bwd", C.c_int), ("row_stride", C.c_int),
                ("f8_sat", C.c_void_p), ("f8_act_scale", C.c_void_p), ("f8_act_amax", C.c_void_p),
                ("fp8_wgrad", C.c_int), ("f8_act_scale_bwd", C.c_void_p), ("fp8_lean", C.c_int)]


_SIGS = {
    "atst_version": (C.c_int, []),
    "atst_tune_gemm_variant": (C.c_int, [C.c_int]),
    "atst_mel_frontend_f32": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "atst_gemm_nt_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                    C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "atst_gemm_nt_resid_ln_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 3 + [C.c_int] + [C.c_void_p] * 7),
    "atst_gemm_nt_lnbwd_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 8 + [C.c_int] + [C.c_void_p] * 4),
    "atst_gemm_nt_resid_ln_fp8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 5 + [C.c_int] + [C.c_void_p] * 11),
    "atst_gemm_nt_lnbwd_q8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 13 + [C.c_int] + [C.c_void_p] * 4),
    "atst_gemm_nt_fp8": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int] * 6 + [C.c_void_p, C.c_int] + [C.c_void_p] * 4 + [C.c_int, C.c_void_p,
                                   C.c_float, C.c_void_p]),
    "atst_quant_fp8_bf16": (C.c_int, [C.c_void_p, C.c_size_t, C.c_float, C.c_void_p, C.c_void_p]),
    "atst_quant_fp8_dyn_bf16": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "atst_fp8_update_scales": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_void_p]),
    "atst_quant_bf16_table_fp8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "atst_quant_weights_fp8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "atst_gemm_tn_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                    C.c_int, C.c_void_p]),
    "atst_gemm_tn_fp8": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int] * 5 + [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "atst_gemm_tn_group_bf16": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "atst_gemm_tn_group_fp8": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "atst_layernorm_fwd": (C.c_int, [C.c_void_p] * 6 + [C.c_int, C.c_int, C.c_void_p]),
    "atst_layernorm_fwd_f32": (C.c_int, [C.c_void_p] * 4 + [C.c_int, C.c_int, C.c_void_p]),
    "atst_layernorm_bwd": (C.c_int, [C.c_void_p] * 9 + [C.c_int] + [C.c_void_p] * 3 + [C.c_int, C.c_int, C.c_void_p]),
    "atst_attention_fwd": (C.c_int, [C.c_void_p] * 4 + [C.c_int] * 3 + [C.c_void_p]),
    "atst_attention_bwd": (C.c_int, [C.c_void_p] * 7 + [C.c_int] * 3 + [C.c_void_p]),
    "atst_attention_fp8_ok": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "atst_attention_fwd_fp8": (C.c_int, [C.c_void_p] * 8 + [C.c_int] * 3 + [C.c_void_p]),
    "atst_attention_bwd_fp8": (C.c_int, [C.c_void_p] * 9 + [C.c_int] * 3 + [C.c_void_p]),
    "atst_patchify_bf16": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "atst_gather_rows_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "atst_scatter_rows_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "atst_colsum_bf16_f32": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "atst_cast_bf16": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "atst_split3_bf16": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "atst_rrc_bicubic_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int] * 5 + [C.c_void_p]),
    "atst_log_mixup_exp_f32": (C.c_int, [C.c_void_p] * 7 + [C.c_int] * 4 + [C.c_void_p]),
    "atst_transpose_bf16_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "atst_transpose_bf16_2d": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "atst_bn_stats_f32": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "atst_bn_finish_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_int, C.c_void_p]),
    "atst_bn_apply_relu_bf16": (C.c_int, [C.c_void_p] * 5 + [C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "atst_bn_apply_relu_split3_bf16": (C.c_int, [C.c_void_p] * 5 + [C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "atst_bn_relu_bwd_sums": (C.c_int, [C.c_void_p] * 6 + [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "atst_bn_bwd_dx_bf16": (C.c_int, [C.c_void_p] * 8 + [C.c_float, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "atst_bn_bwd_dx_f32": (C.c_int, [C.c_void_p] * 8 + [C.c_float, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "atst_byol_loss_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_void_p]),
    "atst_adamw_ema_step": (C.c_int, [C.c_void_p] * 8 + [C.c_size_t, C.c_size_t] + [C.c_double] * 8 + [C.c_void_p]),
    "atst_encoder_ws_bytes": (C.c_size_t, [C.c_int] * 7),
    "atst_encoder_hp_ws_bytes": (C.c_size_t, [C.c_int] * 7),
    "atst_encoder_ws_bytes_geo": (C.c_size_t, [C.c_int] * 9),
    "atst_encoder_hp_fwd": (C.c_int, [C.POINTER(Encoder), C.c_void_p]),
    "atst_encoder_hp_bwd": (C.c_int, [C.POINTER(Encoder), C.c_void_p]),
    "atst_encoder_hp_out": (C.c_void_p, [C.POINTER(Encoder)]),
    "atst_encoder_hp_dout": (C.c_void_p, [C.POINTER(Encoder)]),
    "atst_encoder_hp_block_out": (C.c_void_p, [C.POINTER(Encoder), C.c_int]),
    "atst_encoder_fwd": (C.c_int, [C.POINTER(Encoder), C.c_void_p]),
    "atst_encoder_bwd": (C.c_int, [C.POINTER(Encoder), C.c_void_p]),
    "atst_encoder_bwd_part": (C.c_int, [C.POINTER(Encoder), C.c_int, C.c_int, C.c_void_p]),
    "atst_encoder_bwd_range": (C.c_int, [C.POINTER(Encoder), C.c_int, C.c_int, C.c_void_p]),
    "atst_encoder_out": (C.c_void_p, [C.POINTER(Encoder)]),
    "atst_encoder_dout": (C.c_void_p, [C.POINTER(Encoder)]),
    "atst_encoder_block_out": (C.c_void_p, [C.POINTER(Encoder), C.c_int]),
    "atst_encoder_tokens": (C.c_void_p, [C.POINTER(Encoder)]),
    "atst_profile_enable": (C.c_int, [C.c_int]),
    "atst_profile_kinds": (C.c_int, []),
    "atst_profile_name": (C.c_char_p, [C.c_int]),
    "atst_profile_collect": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_longlong)]),
}

_lib = None


def exported_symbols():
    return sorted(_SIGS)


def load(path: Optional[str] = None):
    """dlopen the library and bind every declared symbol; raises HipError when the .so or a symbol is missing."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise HipError(f"{p} not found -- run `python -c 'import __graft_entry__ as g; g.build()'` (hipcc, gfx950); "
                       "there is no CPU fallback for the product path")
    lib = C.CDLL(p)
    for name, (res, args) in _SIGS.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise HipError(f"{p} does not export {name}") from e
        fn.restype, fn.argtypes = res, args
    if lib.atst_version() != ABI_VERSION:                    # struct layouts / buffer contracts differ between versions: refuse, do not guess
        raise HipError(f"{p} reports ABI version {lib.atst_version()}, these bindings are written for {ABI_VERSION} (include/atst_hip.h) -- rebuild")
    if path is None:
        _lib = lib
        for v in filter(None, os.environ.get("ATST_TUNE", "").split(",")):      # tuning hooks for A/B runs (include/atst_hip.h)
            lib.atst_tune_gemm_variant(int(v))
    return lib


def ptr(t: Optional[torch.Tensor]):
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous(), "HIP kernels take contiguous device tensors"
    return t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


def check(rc: int, what: str = ""):
    if rc != 0:
        raise HipError(f"libatst_hip {what} failed with code {rc}" + (" (ATST_EINVAL: unsupported shape/argument)" if rc == 1001 else ""))


def call(name: str, *args):
    lib = load()
    check(getattr(lib, name)(*args), name)
