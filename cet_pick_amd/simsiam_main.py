"""`python -m cet_pick_amd.simsiam_main simsiam3d --arch simsiam2d_18 --dataset simsiam3d ...` - the reference's
cet_pick/simsiam_main.py (:25-166; the documented exploration command, docs/explore.md:67): SimSiam training of the 2-D /
slice-wise / 2d3d encoders with SGD, per-epoch `adjust_learning_rate`, the reference's checkpoint names and `log.txt` line.
One process per GPU; under torch.distributed.run the ranks shard the crops (DistributedSampler semantics: `set_epoch`,
drop_last) and exchange gradients over RCCL (hipops.GradExchange: one bucketed all-reduce of the flat gradient arena)
with SyncBN statistics.

The reference datasets (MRC lists, torchvision augmentation) are out of scope: the crops come from the DoG picker and
the crop kernels on synthetic tomograms (datasets/synthetic_datasets.py), same batch contract.
"""
import os
import random

import numpy as np
import torch
import torch.distributed as dist

from . import hipops as H
from .datasets.synthetic_datasets import SyntheticSimSiamDataset
from .models.model import create_model, load_model, save_model
from .opts import opts
from .trains.train_factory import train_factory
from .utils.utils import TextLog, adjust_learning_rate


def init_distributed(opt):
    """simsiam_main.py:27-47 / main.py:24-40: WORLD_SIZE decides; LOCAL_RANK names the GPU.  Returns (rank, world)."""
    opt.world_size = int(os.environ.get("WORLD_SIZE", max(opt.world_size, 1)))
    opt.distributed = opt.world_size > 1
    if not opt.distributed:
        opt.gpu = opt.gpus[0] if opt.gpus[0] >= 0 else None
        if opt.gpu is None:
            raise RuntimeError("the MI355X path has no CPU mode (--gpus -1)")
        opt.device = torch.device("cuda", opt.gpu)
        return 0, 1
    opt.gpu = int(os.environ.get("LOCAL_RANK", max(opt.local_rank, 0)))
    backend = os.environ.get("CETPICK_DIST_BACKEND", opt.dist_backend)      # gloo: one-GPU rehearsal of the N>1 path
    if backend == "nccl":
        torch.cuda.set_device(opt.gpu)
    else:
        opt.gpu = opt.gpu % max(torch.cuda.device_count(), 1)
    dist.init_process_group(backend=backend, init_method=opt.dist_url)
    dist.barrier()
    opt.rank = dist.get_rank()
    opt.device = torch.device("cuda", opt.gpu)
    return dist.get_rank(), dist.get_world_size()


def build(opt):
    """Everything `main` sets up before its epoch loop (simsiam_main.py:25-93): process group, model, SGD, checkpoint resume,
    trainer (step engine for the 2-D encoder), dataset.  -> (opt, model, optimizer, trainer, dataset) (+ .rank / .world on opt)."""
    torch.manual_seed(opt.seed)
    rank, world = init_distributed(opt)
    opt.rank_, opt.world_ = rank, world
    Dataset = SyntheticSimSiamDataset
    opt = opts().update_dataset_info_and_set_heads(opt, Dataset)

    print("Creating model...")
    model = create_model(opt.arch, opt.heads, opt.head_conv, local_path=opt.pretrained_model)
    if opt.distributed:
        H.convert_sync_batchnorm(model)
    optimizer = torch.optim.SGD(filter(lambda p: p.requires_grad, model.parameters()), opt.lr)
    opt.start_epoch_ = 0
    if opt.load_model != "":
        model, optimizer, opt.start_epoch_ = load_model(model, opt.load_model, optimizer, opt.resume, opt.lr, opt.lr_step)

    trainer = train_factory[opt.task](opt, model, optimizer)
    if opt.distributed:
        trainer.set_distributed_device(opt.gpu)
    else:
        trainer.set_device(opt.gpus, opt.chunk_sizes, opt.device)

    print("Setting up data...")
    from .datasets.tomo_files import use_files
    if use_files(opt):
        from .datasets.tomo_files import TomoFileSimSiamDataset as Dataset
    dataset = Dataset(opt, "train", (3, opt.bbox, opt.bbox), sigma1=opt.dog, device=opt.device, rank=rank, world=world)
    return opt, model, optimizer, trainer, dataset


def main(opt):
    opt, model, optimizer, trainer, dataset = build(opt)
    rank, start_epoch = opt.rank_, opt.start_epoch_
    logger = TextLog(opt, enabled=rank == 0)
    print("Starting training...")
    for epoch in range(start_epoch + 1, opt.num_epochs + 1):
        np.random.seed(epoch)
        random.seed(epoch)
        dataset.set_epoch(epoch)
        adjust_learning_rate(opt, optimizer, epoch)
        mark = epoch if opt.save_all else "last"
        log_dict_train, _ = trainer.train(epoch, dataset)
        logger.write("epoch: {} |".format(epoch))
        for k, v in log_dict_train.items():
            logger.write("{} {:8f} | ".format(k, v))
        if rank == 0:
            if opt.val_intervals > 0 and epoch % opt.val_intervals == 0:
                save_model(os.path.join(opt.save_dir, "model_{}.pth".format(mark)), epoch, model, optimizer)
            elif not opt.distributed:
                save_model(os.path.join(opt.save_dir, "model_last_contrastive.pth"), epoch, model, optimizer)
        logger.write("\n")
        if epoch in opt.lr_step and rank == 0:
            save_model(os.path.join(opt.save_dir, "model_{}.pth".format(epoch)), epoch, model, optimizer)
    logger.close()
    trainer.close()
    if opt.distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main(opts().parse())
