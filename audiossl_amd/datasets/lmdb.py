"""``LMDBDataset`` -- the reference's waveform store reader (audiossl/datasets/lmdb.py:12-98): same constructor, subset
selection (python ``random`` stream seeded 1234 at import, :8), ``cycle()`` window, item contract
``(transform(waveform[1,N]), label[527])`` (+ key).

Storage back ends.  The reference reads LMDB files whose values were written with ``pyarrow.serialize`` (legacy format,
removed from pyarrow >= 15; scripts/dataset_preprocess/dataset2lmdb.py:16-23,111-145).  Neither ``lmdb`` nor a pyarrow
that still has ``deserialize`` is installed in this image and no reference store ships with the repository, so the
on-disk decoder cannot be pinned here: ``open_store`` uses them when they import (identical to the reference's calls)
and otherwise raises a clear error.  Everything above the key-value layer (what the training loop depends on) is
implemented against the small ``get(key) -> (waveform ndarray [1,1,N], label ndarray [1,C])`` interface and is
covered on CPU with ``DictStore`` (tests/test_host_logic_cpu.py)."""
from __future__ import annotations

import os
import random
from copy import deepcopy

import torch
import torch.utils.data as data

random.seed(1234)                                    # ref: datasets/lmdb.py:8


class DictStore:
    """In-memory key-value store with the two metadata entries of the reference layout (``__keys__``, ``__len__``)."""

    def __init__(self, items: dict):
        self.items = dict(items)
        self.keys = list(self.items)

    def get(self, key):
        return self.items[key]

    def __len__(self):
        return len(self.keys)


class _LmdbStore:
    """The reference's own calls (lmdb.open readonly / pa.deserialize), used only where both libraries exist."""

    def __init__(self, path):
        import lmdb
        import pyarrow as pa
        if not hasattr(pa, "deserialize"):
            raise ImportError("this pyarrow has no legacy pa.deserialize (removed in pyarrow 15)")
        self._de = pa.deserialize
        self.env = lmdb.open(path, subdir=os.path.isdir(path), readonly=True, lock=False, readahead=False, meminit=False)
        with self.env.begin(write=False) as txn:
            self._len = self._de(txn.get(b"__len__"))
            self.keys = self._de(txn.get(b"__keys__"))
        self.txn = self.env.begin(write=False)

    def get(self, key):
        return self._de(self.txn.get(key))

    def __len__(self):
        return self._len


def open_store(path):
    try:
        return _LmdbStore(path)
    except ImportError as e:
        raise RuntimeError(f"cannot open {path}: the reference's LMDB + legacy-pyarrow store needs `lmdb` and a pyarrow with "
                           f"pa.deserialize ({e}); pass store=DictStore(...) or any object with .keys / .get(key)") from e


class LMDBDataset(data.Dataset):
    def __init__(self, db_path, split, subset=None, transform=None, target_transform=None, return_key=False, store=None):
        self.db_path, self.return_key, self.subset = db_path, return_key, subset
        name = {"train": "train.lmdb", "valid": "valid.lmdb"}.get(split, "eval.lmdb")           # ref: lmdb.py:16-21
        self.store = store if store is not None else open_store(os.path.join(db_path, name))
        self.length = len(self.store)
        self.keys = list(self.store.keys)
        self.org_keys = deepcopy(self.keys)
        self.start = 0
        if subset is not None and subset < self.length:                                          # ref: lmdb.py:33-38
            self.length = subset
            random.shuffle(self.keys)
            self.org_keys = deepcopy(self.keys)
            self.keys = self.keys[:subset]
            self.start = subset
        self.transform, self.target_transform = transform, target_transform
        self.num_classes = self.store.get(self.keys[0])[1].shape[1]
        self.sr = 16000

    def __getitem__(self, index):
        key = self.keys[index]
        unpacked = self.store.get(key)
        waveform, label = torch.from_numpy(unpacked[0]).squeeze(0), torch.from_numpy(unpacked[1]).squeeze(0)
        if self.transform is not None:
            out = self.transform(waveform)
            if self.target_transform is not None:
                out = list(out)
                out[0], label = self.target_transform(out[0], label)
                out = tuple(out)
        else:
            out = waveform
        return (out, label, key) if self.return_key else (out, label)

    def cycle(self):
        """Slide the subset window over the shuffled key list (ref: lmdb.py:82-90)."""
        if self.start + self.subset > len(self.org_keys):
            self.keys = self.org_keys[self.start:] + self.org_keys[:self.start + self.subset - len(self.org_keys)]
            random.shuffle(self.org_keys)
            self.start = 0
        else:
            self.keys = self.org_keys[self.start:self.start + self.subset]
            self.start = self.start + self.subset

    def __len__(self):
        return len(self.keys)

    def __repr__(self):
        return self.__class__.__name__ + " (" + str(self.db_path) + ")"
