"""CLI entry with the reference's arguments (audiossl/methods/atst/train.py:11-49): lr linear scaling, per-GPU batch,
auto-resume from ``last.ckpt``.  One process per GPU: launch with ``python -m torch.distributed.run --nproc-per-node N``."""
import os
from argparse import ArgumentParser

import torch
import torch.distributed as dist

from ...trainer import Trainer
from .data import ATSTDataModule
from .model import ATSTLightningModule
from .transform import ATSTBatchViews


def main(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and not dist.is_initialized():
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group("nccl")
    args.nproc = world                                       # the linear-scaling rule uses the devices actually training (ref: nproc = Trainer devices)
    args.learning_rate = args.learning_rate * args.nproc * args.batch_size_per_gpu / 256        # ref: train.py:12
    dict_args = vars(args)
    model = ATSTLightningModule(**dict_args)
    data = ATSTDataModule(**dict_args)
    trainer = Trainer(max_steps=args.max_steps, default_root_dir=args.save_path, every_n_epochs=20,
                      batch_hook=ATSTBatchViews())
    last_ckpt = os.path.join(args.save_path, "last.ckpt") if args.save_path else None
    trainer.fit(model, datamodule=data, ckpt_path=last_ckpt if last_ckpt and os.path.exists(last_ckpt) else None)


if __name__ == "__main__":
    parser = ArgumentParser("ATST")
    parser.add_argument("--save_path", type=str)
    parser.add_argument("--nproc", type=int, default=1)
    parser = ATSTLightningModule.add_model_specific_args(parser)
    parser = ATSTDataModule.add_data_specific_args(parser)
    main(parser.parse_args())
