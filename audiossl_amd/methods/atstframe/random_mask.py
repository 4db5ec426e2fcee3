"""Mask samplers of the ATST-Frame transform (host side, numpy / torch CPU RNG like the reference).

``get_mask`` restates ``fairseq.data.data_utils.compute_mask_indices`` for the one way the reference calls it
(audiossl/methods/atstframe/random_mask.py:5-15 from transform.py:88-89: shape (1, S), mask_type "static",
no_overlap False, min_space 0, min_masks 2); fairseq is a third-party dependency that is absent from /root/reference,
so this follows fairseq 0.12's published algorithm and is "parity unpinned" at that boundary.
``get_mask_one`` / ``get_mask_batch`` restate random_mask.py:27-36."""
import numpy as np
import torch
import torch.nn.functional as F


def block_mask(num_patches=250, mask_prob=0.65, mask_length=5, min_masks=2, rng=None, mask_type="static", mask_other=0):
    """-> bool [num_patches]; `num` spans at distinct random starts (spans may overlap).  "static": every span is
    `mask_length` long; "uniform": span lengths ~ randint(mask_other, 2*mask_length + 1) (fairseq 0.12
    compute_mask_indices, no_overlap=False, bsz 1 -- SURVEY.md Appendix A.3).  Draw order on the numpy stream as in
    fairseq: rand() for the span count, [randint for the lengths,] choice for the starts."""
    rng = np.random if rng is None else rng
    S = num_patches
    num = max(min_masks, int(mask_prob * S / float(mask_length) + rng.rand()))
    if mask_type == "static":
        lengths = np.full(num, mask_length)
    elif mask_type == "uniform":
        lengths = rng.randint(mask_other, mask_length * 2 + 1, size=num)
    else:
        raise NotImplementedError(f"mask_type {mask_type!r}: the reference transform only asks for static / uniform")
    if lengths.sum() == 0:
        lengths[0] = min(mask_length, S - 1)
    min_len = int(lengths.min())
    if S - min_len <= num:
        min_len = S - num - 1
    starts = rng.choice(S - min_len, num, replace=False)
    idx = np.concatenate([starts[j] + np.arange(lengths[j]) for j in range(num)]) if num else np.zeros(0, dtype=np.int64)
    mask = np.zeros(S, dtype=bool)
    mask[np.unique(idx[idx < S])] = True
    return mask


def get_mask(batch_size, num_patches, mask_ratio, padding_mask=None, no_overlap=False, min_length=5, type="static", other=0):
    """ref: random_mask.py:5-15.  bsz > 1 additionally equalises the masked count to the batch minimum (fairseq's
    require_same_masks) -- the transform always calls with bsz 1 (transform.py:88-91), where that step is a no-op."""
    if no_overlap or padding_mask is not None:
        raise NotImplementedError("only the overlapping block mask without padding (what the reference transform uses) is restated")
    rows = [block_mask(num_patches, mask_ratio, min_length, mask_type=type, mask_other=other) for _ in range(batch_size)]
    if batch_size > 1:
        n_min = min(int(r.sum()) for r in rows)
        for r in rows:
            on = np.flatnonzero(r)
            if len(on) > n_min:
                r[:] = False
                r[np.random.choice(on, n_min, replace=False)] = True
    return torch.from_numpy(np.stack(rows))


def get_mask_one(num_patches, available_patches, mask_ratio):
    m = torch.randperm(available_patches) < available_patches * mask_ratio
    return F.pad(m, (0, num_patches - available_patches), value=True)


def get_mask_batch(batch_size, num_patches, mask_ratio):
    return torch.stack([torch.randperm(num_patches) < num_patches * mask_ratio for _ in range(batch_size)])
