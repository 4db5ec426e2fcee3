"""Schedules and optimizer parameter grouping -- same names and semantics as audiossl/utils/common.py:3-80."""
import argparse

import numpy as np


def concat_all_gather(tensor):
    """all_gather + cat over ranks, no gradient. ref: utils/common.py:3-14."""
    import torch
    with torch.no_grad():
        parts = [torch.ones_like(tensor) for _ in range(torch.distributed.get_world_size())]
        torch.distributed.all_gather(parts, tensor, async_op=False)
        return torch.cat(parts, dim=0)


def cosine_scheduler_epoch(base_value, final_value, epochs, niter_per_ep, warmup_epochs=0, start_warmup_value=0):
    """Per-iteration table over whole epochs (the downstream heads' schedule). ref: utils/common.py:16-27."""
    warm = warmup_epochs * niter_per_ep
    head = np.linspace(start_warmup_value, base_value, warm) if warmup_epochs > 0 else np.array([])
    n = epochs * niter_per_ep - warm
    tail = final_value + 0.5 * (base_value - final_value) * (1 + np.cos(np.pi * np.arange(n) / n))
    table = np.concatenate((head, tail))
    assert len(table) == epochs * niter_per_ep
    return table


def cosine_scheduler_step(base_value, final_value, max_steps, warmup_steps=0, start_warmup_value=0):
    """Per-step table: linear warm-up to base_value, then half-cosine to final_value. ref: utils/common.py:29-39."""
    head = np.linspace(start_warmup_value, base_value, warmup_steps) if warmup_steps > 0 else np.array([])
    n = max_steps - warmup_steps
    tail = final_value + 0.5 * (base_value - final_value) * (1 + np.cos(np.pi * np.arange(n) / n))
    table = np.concatenate((head, tail))
    assert len(table) == max_steps
    return table


def get_params_groups(model, no_weight_decay_attr: list = [], debug=False):
    """Two AdamW groups: [regularised] and [biases + every 1-D tensor, weight_decay 0]. ref: utils/common.py:41-68."""
    reg, noreg, reg_n, noreg_n = [], [], [], []
    for name, p in model.named_parameters():
        if not p.requires_grad:
            continue
        flagged = any(getattr(p, a, False) for a in no_weight_decay_attr)
        if name.endswith(".bias") or p.ndim == 1 or flagged:
            noreg.append(p); noreg_n.append(name)
        else:
            reg.append(p); reg_n.append(name)
    if debug:
        return reg_n, noreg_n
    return [{"params": reg}, {"params": noreg, "weight_decay": 0.0}]


def bool_flag(s):
    """ref: utils/common.py:69-80."""
    v = s.lower()
    if v in {"off", "false", "0"}:
        return False
    if v in {"on", "true", "1"}:
        return True
    raise argparse.ArgumentTypeError("invalid value for a boolean flag")
