"""``ATSTDataModule`` with the reference's constructor / argparse surface (audiossl/methods/atst/data.py:6-42).
``data_path`` opens the reference's LMDB store through audiossl_amd.datasets.LMDBDataset (needs `lmdb` + legacy pyarrow,
see that module); any map-style dataset returning ``(waveform[1,N], label)`` can be passed as ``dataset=``; without
either a synthetic AudioSet-shaped dataset is used (N(0, 0.1^2) noise, 10 s)."""
from __future__ import annotations

import torch
from torch.utils import data

from .transform import ATSTTrainTransform


class SyntheticWaveDataset(data.Dataset):
    def __init__(self, n_items=2048, seconds=10.0, sr=16000, transform=None, seed=1234):
        self.n, self.len, self.transform, self.seed = n_items, int(seconds * sr), transform, seed

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        g = torch.Generator().manual_seed(self.seed + i)
        wave = torch.clamp(0.1 * torch.randn(1, self.len, generator=g), -1.0, 1.0)
        label = torch.zeros(1, 527)
        return (self.transform(wave) if self.transform else wave), label


class ATSTDataModule:
    def __init__(self, data_path=None, batch_size_per_gpu=256, num_workers=10, subset=200000, train_len=6.0, dataset=None,
                 **kwargs):
        self.transform = ATSTTrainTransform(anchor_len=(train_len, train_len), positive_len=(train_len, train_len))
        if dataset is None and data_path is not None:        # ref: data.py:18-23 (LMDBDataset(data_path, split="train", subset=...))
            from ...datasets import LMDBDataset
            dataset = LMDBDataset(data_path, split="train", subset=subset, transform=self.transform)
        if dataset is None:
            dataset = SyntheticWaveDataset(min(subset, 4096), seconds=max(train_len, 10.0), transform=self.transform)
        self.dataset = dataset
        self.batch_size, self.num_workers = batch_size_per_gpu, num_workers
        self.hparams = dict(data_path=data_path, batch_size_per_gpu=batch_size_per_gpu, num_workers=num_workers,
                            subset=subset, train_len=train_len)

    def train_dataloader(self, rank=0, world=1):
        sampler = None
        if world > 1:                                  # what Lightning injects: non-shuffling DistributedSampler
            sampler = data.distributed.DistributedSampler(self.dataset, world, rank, shuffle=False, drop_last=True)
        return data.DataLoader(self.dataset, batch_size=self.batch_size, num_workers=self.num_workers, sampler=sampler,
                               drop_last=True)

    @staticmethod
    def add_data_specific_args(parent_parser):
        parser = parent_parser.add_argument_group("ATSTData")
        parser.add_argument("--data_path", type=str, default=None, help="data path")
        parser.add_argument("--batch_size_per_gpu", default=256, type=int, help="distinct samples loaded on one GPU")
        parser.add_argument("--num_workers", default=10, type=int, help="data loading workers per GPU")
        parser.add_argument("--subset", default=200000, type=int, help="subset of training data")
        parser.add_argument("--train_len", default=6.0, type=float, help="length of training segment")
        return parent_parser
