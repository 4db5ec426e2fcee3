"""``FrameATSTTrainTransform`` -- same constructor and call contract as audiossl/methods/atstframe/transform.py:14-101.

Reference, per item on a CPU worker: ONE random crop of the waveform -> log-mel (win_length 640 in the shipped recipe)
-> view 0 = positive_transform1(mel), view 1 = positive_transform2(mel) (each Mixup + frequency-only RandomResizeCrop or
identity, by aug_tea / aug_stu) -> ONE mask shared by both views -> ``(crops, lengths, masks)``.

Here the mel front end and the augmentations are HIP kernels and forked workers must not touch the GPU, so the work is
split like methods/atst/transform.py:
  * per item, on the worker (``__call__`` with a CPU waveform): the random crop and the mask draw (numpy / torch CPU
    RNGs, as in the reference) -> ``([crop[1, n]], [len, len], [mask, mask])`` -- the crop is collated once;
  * per batch, on the GPU (``FrameATSTBatchViews``): log-mel once, then the two views.
Called with a CUDA waveform the transform runs both stages and returns mel crops ``[1, 64, T]`` like the reference."""
from __future__ import annotations

import torch
import torch.nn.functional as F

from ...frontend import LogMelFrontend
from ...transforms import BatchMixup, BatchRandomResizeCrop, RandomCrop
from . import random_mask


def get_num_patches(height=64, width=1001, patch_height=16, patch_width=16):
    """ref: audiossl/models/atst/audio_transformer.py:367-371."""
    return (height // patch_height) * (width // patch_width)


class FrameATSTBatchViews:
    """GPU stage: the collated crop [B,1,n] -> [view0, view1], each [B,1,64,T].  Mixup / RandomResizeCrop objects are
    per view like the reference's two Compose pipelines (transform.py:46-66): each Mixup keeps its own memory bank."""

    def __init__(self, win_length=1024, aug_tea=True, aug_stu=True, mix_up=True, freq_wrap=True, device=None, sr=16000, n_mels=64):
        self.mel_feature = LogMelFrontend(win_length, device, sr=sr, n_mels=n_mels)             # ref: transform.py:15-16 (sr, n_mels)
        def pipeline(on):
            if not on:
                return (None, None)
            return (BatchMixup() if mix_up else None,
                    BatchRandomResizeCrop((1, 1.0), time_scale=(1.0, 1.0)) if freq_wrap else None)
        self.pipes = [pipeline(aug_tea), pipeline(aug_stu)]

    def __call__(self, waves, lengths=None):
        w = waves[0] if isinstance(waves, (list, tuple)) else waves
        mel = self.mel_feature(w)
        views = []
        for mix, rrc in self.pipes:
            m = mel
            if mix is not None:
                m = mix(m)
            if rrc is not None:
                m = rrc(m)
            views.append(m)
        return views


class FrameATSTTrainTransform:
    def __init__(self, sr=16000, win_length=1024, aug_tea=True, aug_stu=True, mix_up=True, freq_wrap=True, mask_ratio=0.75,
                 mask_nooverlap=False, min_mask_len=2, mask_len=5, mask_type="random", anchor_len=6., patch_h=64, patch_w=4,
                 n_mels=64, **kwargs):
        # sr / n_mels / patch_h / patch_w as the reference threads them (transform.py:14-17, train.py:15,50-51: spec_h = n_mels).  The HIP front
        # end is compiled for 64 or 128 bands and the engine for ONE patch row (patch_h = n_mels) of 4 or 8 frames -- BASELINE.json configs[4]
        # is sr 32000, n_mels 128, patch 128 x 8.
        if n_mels not in (64, 128) or patch_h != n_mels or patch_w not in (4, 8):
            raise NotImplementedError(f"supported geometries: n_mels 64 / 128 with patch_h = n_mels (one patch row) and patch_w 4 / 8; got "
                                      f"n_mels {n_mels}, patch {patch_h} x {patch_w}")
        self.sr = sr
        self.anchor_len = anchor_len
        self.max_positive_len = self.anchor_len
        self.mask_ratio, self.mask_type = mask_ratio, mask_type
        self.aug_tea, self.aug_stu, self.mix_up, self.freq_wrap = aug_tea, aug_stu, mix_up, freq_wrap
        self.patch_h, self.patch_w, self.mask_len, self.n_mels = patch_h, patch_w, mask_len, n_mels
        self.mask_nooverlap, self.min_mask_len = mask_nooverlap, min_mask_len
        self.win_length = win_length
        self._crop = RandomCrop(16000 * 6)
        self._gpu = None

    def batch_views(self, device=None) -> FrameATSTBatchViews:
        """The GPU stage configured like this transform (what Trainer(batch_hook=...) wants)."""
        return FrameATSTBatchViews(self.win_length, self.aug_tea, self.aug_stu, self.mix_up, self.freq_wrap, device, sr=self.sr, n_mels=self.n_mels)

    def _mask(self, num_patches):
        # ref: transform.py:86-91
        if self.mask_type == "random":
            return random_mask.get_mask_one(num_patches, num_patches, self.mask_ratio)
        kind = "static" if self.mask_type == "block" else "uniform"
        return random_mask.get_mask(1, num_patches, self.mask_ratio, no_overlap=self.mask_nooverlap, min_length=self.mask_len,
                                    type=kind, other=self.min_mask_len).squeeze(0)

    def __call__(self, input):
        anchor_len = self.anchor_len
        n = int(anchor_len * 16000)            # the reference counts anchor_len in units of 16000 SAMPLES whatever `sr` is (transform.py:76,81,93-98):
                                               # 10 s at 32 kHz = anchor_len 20 -> 320000 samples -> 2001 frames; `sr` only enters the mel filterbank
        self._crop.size = n
        crop = self._crop(input)                                                     # ref: transform.py:76-78
        frames = n // 160 + 1
        mask = self._mask(get_num_patches(self.n_mels, frames, self.patch_h, self.patch_w))
        lengths = [frames, frames]                                                   # ref: transform.py:95,98
        masks = [mask, mask]                                                         # one mask shared by both views (:99)
        if not input.is_cuda:
            return [crop], lengths, masks
        if self._gpu is None:
            self._gpu = self.batch_views(input.device)
        views = self._gpu([crop.unsqueeze(0)])
        pad = int((self.max_positive_len * 16000) // 160 - n // 160)                # ref: transform.py:93-97 (0 for this recipe)
        return [F.pad(v[0], (0, pad)) for v in views], lengths, masks
