"""Scene / timestamp embeddings of a pre-trained ATST-Frame encoder -- the reference's
``audiossl/methods/atstframe/embedding.py:19-127`` (``load_model``, ``get_scene_embedding``,
``get_timestamp_embedding``) on the HIP path: log-mel front end (win_length 1024, as the reference's module-level
``melspec_t``), 10 s chunks (1001 frames, the length of the positional table), all 12 blocks' ``norm_frame`` outputs."""
from __future__ import annotations

import torch

from ...frontend import LogMelFrontend
from .model import FrameATSTLightningModule

N_BLOCKS = 12
CHUNK_LEN = 1001            # 10 seconds: consistent with the length of the positional embedding (embedding.py:61,107)


def load_model(model_path):
    """ref: embedding.py:19-38.  Returns ``model.teacher.encoder`` with the attributes the HEAR-style callers read."""
    ckpt = torch.load(model_path, map_location="cpu", weights_only=False)
    module = FrameATSTLightningModule.load_from_checkpoint(model_path)
    enc = module.model.teacher.encoder
    enc.hyper_param = ckpt.get("hyper_parameters", {})
    enc.sample_rate = 16000
    enc.scene_embedding_size = enc.embed_dim * 2 * N_BLOCKS      # as in the reference (embedding.py:30), though n_blocks * C is returned
    enc.timestamp_embedding_size = enc.embed_dim * N_BLOCKS
    enc.transform = LogMelFrontend(1024)
    enc._owner = module                                          # keep the engine alive
    return enc


def _mel(audio, model):
    if audio.dim() == 2:
        audio = audio.unsqueeze(1)
    assert audio.dim() == 3
    return model.transform(audio)                                # [B, 1, 64, T]


def _chunks(total_len):
    for i in range(total_len // CHUNK_LEN + 1):
        start, end = i * CHUNK_LEN, min((i + 1) * CHUNK_LEN, total_len)
        if end > start:
            yield start, end


def get_scene_embedding(audio, model):
    """[B, N_BLOCKS * C]: mean over 10 s chunks of the masked frame means of all blocks. ref: embedding.py:41-82."""
    mel = _mel(audio, model)
    out = []
    for start, end in _chunks(mel.shape[-1]):
        chunk = mel[..., start:end]
        length = torch.full((mel.shape[0],), chunk.shape[-1], dtype=torch.int64)
        out.append(model.get_intermediate_layers(chunk, length, n=N_BLOCKS))
    return torch.stack(out, dim=0).mean(dim=0)


def get_timestamp_embedding(audio, model):
    """([B, T, N_BLOCKS * C], timestamps [B, T] in ms, 40 ms per frame). ref: embedding.py:85-127."""
    mel = _mel(audio, model)
    out = []
    for start, end in _chunks(mel.shape[-1]):
        chunk = mel[..., start:end]
        length = torch.full((mel.shape[0],), chunk.shape[-1], dtype=torch.int64)
        out.append(model.get_intermediate_layers(chunk, length, n=N_BLOCKS, scene=False))
    emb = torch.cat(out, dim=1)
    ts = (torch.arange(emb.shape[1]) * 40).float().unsqueeze(0).expand(mel.shape[0], -1)
    return emb, ts
