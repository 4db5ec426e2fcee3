from .lmdb import LMDBDataset, DictStore, open_store  # noqa: F401
