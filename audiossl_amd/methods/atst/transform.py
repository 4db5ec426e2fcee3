"""``ATSTTrainTransform`` -- same constructor and call contract as audiossl/methods/atst/transform.py:12-74.

The reference computes the mel features on CPU dataloader workers (torchaudio).  Here the mel front end is a HIP kernel,
and forked workers must not touch the GPU, so the transform works in two stages:
  * per item, on the worker (``__call__``): random crops of the waveform, exactly like the reference's RandomCrop; it
    returns ``(crops, lengths)`` with waveform crops ``[1, n]`` zero-padded to the longest view and the frame lengths
    ``n // 160 + 1`` the reference reports (transform.py:62,68-73);
  * per batch, on the GPU (``ATSTBatchViews``): log-mel (HIP) -> Mixup -> RandomResizeCrop -> pad, producing the
    ``melspecs`` list of ``[B,1,64,T]`` tensors that ``training_step`` consumes.
Calling the transform with a CUDA tensor runs both stages at once and returns mel crops like the reference."""
from __future__ import annotations

import random

import torch
import torch.nn.functional as F

from ...frontend import LogMelFrontend
from ...transforms import BatchMixup, BatchRandomResizeCrop, RandomCrop

random.seed(1234)                                   # ref: transform.py:8


class ATSTBatchViews:
    """GPU stage: waveform views [B,1,n_v] (+ lengths) -> list of normalised log-mel views [B,1,64,T]."""

    def __init__(self, virtual_crop=1.5, win_length=1024, mix_up=True, resize_crop=True, freq_only=False, device=None):
        self.mel_feature = LogMelFrontend(win_length, device)
        self.mixup = [BatchMixup() if mix_up else None for _ in range(2)]
        rrc = (lambda: BatchRandomResizeCrop((1, 1.0), time_scale=(1.0, 1.0))) if freq_only else \
              (lambda: BatchRandomResizeCrop((1, virtual_crop)))
        self.rrc = [rrc() if resize_crop else None for _ in range(2)]

    def _view(self, k, w, n):
        m = self.mel_feature(w[..., :n] if n > 0 else w)
        if self.mixup[k] is not None:
            m = self.mixup[k](m)
        if self.rrc[k] is not None:
            m = self.rrc[k](m)
        return m

    def __call__(self, waves, lengths):
        """Items of one view may have different crop lengths (anchor_len[0] != anchor_len[1]): the reference computes mel,
        Mixup and RandomResizeCrop over each item's own crop (transform.py:52-73), so the batch is processed in groups
        of equal frame count and each group is padded back to the common width.  A crop of `frames` frames is taken as
        (frames - 1) * 160 samples: exact whenever the crop length is a multiple of the hop (the shipped 6 s / 10 s
        recipes), up to 159 tail samples short otherwise."""
        mels = []
        t_max = max(int(w.shape[-1]) // 160 for w in waves)
        for v, (w, ln) in enumerate(zip(waves, lengths)):
            k = min(v, 1)
            ln = torch.as_tensor(ln).reshape(-1).cpu()
            uniq = torch.unique(ln).tolist()
            if len(uniq) == 1:
                m = self._view(k, w, (int(uniq[0]) - 1) * 160)
                mels.append(F.pad(m, (0, max(0, t_max + 1 - m.shape[-1]))))
                continue
            out = torch.zeros(w.shape[0], 1, 64, t_max + 1, device=w.device)
            for u in uniq:
                idx = (ln == u).nonzero(as_tuple=True)[0].to(w.device)
                m = self._view(k, w.index_select(0, idx), (int(u) - 1) * 160)
                out.index_copy_(0, idx, F.pad(m, (0, max(0, t_max + 1 - m.shape[-1]))))
            mels.append(out)
        return mels


class ATSTTrainTransform:
    def __init__(self, sr=16000, mask_ratio=0.75, different_positive=True, anchor_len=(6., 6.), positive_len=(6., 6.),
                 virtual_crop=1.5):
        self.different_positive = different_positive
        self.anchor_len, self.positive_len = anchor_len, positive_len
        self.max_positive_len = max(self.positive_len + self.anchor_len)
        self.virtual_crop = virtual_crop
        self._crop = RandomCrop(16000 * 6)
        self._gpu = None

    def _crops(self, wave):
        anchor_len = random.uniform(self.anchor_len[0], self.anchor_len[1])
        self._crop.size = int(anchor_len * 16000)
        c1 = self._crop(wave)
        if self.different_positive:
            positive_len = random.uniform(self.positive_len[0], self.positive_len[1])
            self._crop.size = int(positive_len * 16000)
            c2 = self._crop(wave)
        else:
            positive_len, c2 = anchor_len, c1
        n_max = int(self.max_positive_len * 16000)
        crops = [F.pad(c, (0, n_max - c.shape[-1])) for c in (c1, c2)]
        lengths = [int(anchor_len * 16000) // 160 + 1, int(positive_len * 16000) // 160 + 1]
        return crops, lengths

    def __call__(self, input):
        crops, lengths = self._crops(input)
        if not input.is_cuda:
            return crops, lengths                      # waveform views; the mel stage runs after collate on the GPU
        if self._gpu is None:
            self._gpu = ATSTBatchViews(self.virtual_crop, device=input.device)
        mels = self._gpu([c.unsqueeze(0) for c in crops], lengths)
        return [m[0] for m in mels], lengths
