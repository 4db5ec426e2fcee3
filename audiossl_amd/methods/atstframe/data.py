"""``FrameATSTDataModule`` with the reference's constructor / argparse surface (audiossl/methods/atstframe/data.py:20-111).
Like the reference it does NOT forward ``mix_up`` to the transform (data.py:46-57), so Mixup follows the transform's
default (on) in every augmented branch.  Any map-style dataset returning ``(waveform[1,N], label)`` can be passed as
``dataset=`` (e.g. audiossl_amd.datasets.LMDBDataset); without one a synthetic AudioSet-shaped dataset is used."""
from __future__ import annotations

from torch.utils import data

from ...utils.common import bool_flag
from ..atst.data import SyntheticWaveDataset
from .transform import FrameATSTTrainTransform


class FrameATSTDataModule:
    def __init__(self, data_path=None, batch_size_per_gpu=256, num_workers=10, subset=200000, win_length=1024, aug_tea=True,
                 aug_stu=True, freq_wrap=True, mix_up=True, mask_ratio=0.75, mask_type="block", anchor_len=6., mask_len=5,
                 min_mask_len=2, n_mels=64, dataset=None, **kwargs):
        tkw = {k: kwargs[k] for k in ("patch_h", "patch_w", "mask_nooverlap", "sr") if k in kwargs}
        self.transform = FrameATSTTrainTransform(win_length=win_length, aug_tea=aug_tea, aug_stu=aug_stu, freq_wrap=freq_wrap,
                                                 mask_ratio=mask_ratio, anchor_len=anchor_len, mask_type=mask_type,
                                                 mask_len=mask_len, min_mask_len=min_mask_len, n_mels=n_mels, **tkw)
        if dataset is None:
            if data_path is not None:
                from ...datasets import LMDBDataset
                dataset = LMDBDataset(data_path, split="train", subset=subset, transform=self.transform)
            else:
                dataset = SyntheticWaveDataset(min(subset, 4096), seconds=max(float(anchor_len), 10.0), transform=self.transform)
        self.dataset = dataset
        self.batch_size, self.num_workers = batch_size_per_gpu, num_workers
        self.hparams = dict(data_path=data_path, batch_size_per_gpu=batch_size_per_gpu, num_workers=num_workers, subset=subset,
                            win_length=win_length, aug_tea=aug_tea, aug_stu=aug_stu, freq_wrap=freq_wrap, mix_up=mix_up,
                            mask_ratio=mask_ratio, mask_type=mask_type, anchor_len=anchor_len, mask_len=mask_len,
                            min_mask_len=min_mask_len, n_mels=n_mels)

    def train_dataloader(self, rank=0, world=1):
        sampler = None
        if world > 1:                                  # what Lightning injects: non-shuffling DistributedSampler
            sampler = data.distributed.DistributedSampler(self.dataset, world, rank, shuffle=False, drop_last=True)
        return data.DataLoader(self.dataset, batch_size=self.batch_size, num_workers=self.num_workers, sampler=sampler,
                               drop_last=True)

    @staticmethod
    def add_data_specific_args(parent_parser):
        parser = parent_parser.add_argument_group("FrameATSTData")
        parser.add_argument("--data_path", type=str, default=None, help="data path")
        parser.add_argument("--batch_size_per_gpu", default=256, type=int, help="distinct samples loaded on one GPU")
        parser.add_argument("--num_workers", default=10, type=int, help="data loading workers per GPU")
        parser.add_argument("--subset", default=200000, type=int, help="subset of training data")
        parser.add_argument("--win_length", default=1024, type=int, help="window length")
        parser.add_argument("--aug_tea", default=True, type=bool_flag, help="augment the first view")
        parser.add_argument("--aug_stu", default=True, type=bool_flag, help="augment the second view")
        parser.add_argument("--freq_wrap", default=True, type=bool_flag, help="freq warping or not")
        parser.add_argument("--mix_up", default=True, type=bool_flag, help="mixup or not")
        parser.add_argument("--anchor_len", default=6., type=float, help="length of training samples")
        parser.add_argument("--mask_ratio", default=0.75, type=float, help="masking ratio")
        parser.add_argument("--mask_len", default=5, type=int, help="masking block length")
        parser.add_argument("--min_mask_len", default=2, type=int, help="minimum masking block length")
        parser.add_argument("--n_mels", default=64, type=int, help="number of mel channels")
        parser.add_argument("--mask_type", default="block", type=str, help="masking type: random or block")
        return parent_parser
