"""Mask samplers of the ATST-Frame transform (host side, numpy / torch CPU RNG like the reference).

``get_mask`` restates ``fairseq.data.data_utils.compute_mask_indices`` for the one way the reference calls it
(audiossl/methods/atstframe/random_mask.py:5-15 from transform.py:88-89: shape (1, S), mask_type "static",
no_overlap False, min_space 0, min_masks 2); fairseq is a third-party dependency that is absent from /root/reference,
so this follows fairseq 0.12's published algorithm and is "parity unpinned" at that boundary.
``get_mask_one`` / ``get_mask_batch`` restate random_mask.py:27-36."""
import numpy as np
import torch
import torch.nn.functional as F


def block_mask(num_patches=250, mask_prob=0.65, mask_length=5, min_masks=2, rng=None):
    """-> bool [num_patches]; spans of `mask_length` at `num` distinct random starts (spans may overlap)."""
    rng = np.random if rng is None else rng
    S = num_patches
    num = max(min_masks, int(mask_prob * S / float(mask_length) + rng.rand()))
    span = mask_length
    if S - span <= num:
        span = S - num - 1
    starts = rng.choice(S - span, num, replace=False)
    idx = (starts[:, None] + np.arange(mask_length)[None, :]).reshape(-1)
    mask = np.zeros(S, dtype=bool)
    mask[np.unique(idx[idx < S])] = True
    return mask


def get_mask(batch_size, num_patches, mask_ratio, padding_mask=None, no_overlap=False, min_length=5, type="static", other=0):
    if type != "static" or no_overlap or padding_mask is not None:
        raise NotImplementedError("only the shipped recipe's static, overlapping block mask is restated")
    return torch.from_numpy(np.stack([block_mask(num_patches, mask_ratio, min_length) for _ in range(batch_size)]))


def get_mask_one(num_patches, available_patches, mask_ratio):
    m = torch.randperm(available_patches) < available_patches * mask_ratio
    return F.pad(m, (0, num_patches - available_patches), value=True)


def get_mask_batch(batch_size, num_patches, mask_ratio):
    return torch.stack([torch.randperm(num_patches) < num_patches * mask_ratio for _ in range(batch_size)])
