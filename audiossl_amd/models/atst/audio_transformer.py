"""Upstream names of the encoder classes (audiossl/models/atst/audio_transformer.py:77,367-374).

``AST`` is the type of ``model.{student,teacher}.encoder`` (annotations and isinstance checks of the downstream harness,
audiossl/methods/atst/downstream/model.py:20); ``AST_small()`` / ``AST_base()`` build a stand-alone HIP-backed encoder with
the reference's parameter names for the harness's legacy-checkpoint branch (downstream/train_freeze.py:36-48), which then
calls ``load_state_dict`` on it and uses the inference API (``get_intermediate_layers_chunks``, ``embed_dim``)."""
from .atst import ATST, EncoderView

AST = EncoderView


def _encoder(arch, **kwargs):
    model = ATST(arch=arch, **kwargs)
    enc = model.teacher.encoder
    enc._owner = [model]                 # the encoder's parameters are views into the model's flat buffers: keep it alive
    return enc


def AST_small(patch_h=64, patch_w=4, **kwargs):
    """ref: audio_transformer.py:367-370 (embed_dim 384, depth 12, 6 heads; patch_h x patch_w patches, shipped 64 x 4).  The HIP
    engine takes one patch row of 64 / 128 mel bands x 4 / 8 frames (AtstEngine raises for anything else)."""
    return _encoder("small", patch_h=patch_h, patch_w=patch_w, **kwargs)


def AST_base(patch_h=64, patch_w=4, **kwargs):
    """ref: audio_transformer.py:371-374 (embed_dim 768, depth 12, 12 heads); configs[4] uses 128 x 8 on 32 kHz / 128-mel input."""
    return _encoder("base", patch_h=patch_h, patch_w=patch_w, **kwargs)
