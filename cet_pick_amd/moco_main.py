"""`python -m cet_pick_amd.moco_main moco --arch moco3d_18 ...` - the intended behaviour of the
reference's cet_pick/moco_main.py (:27-207), with the defects listed in SURVEY.md §3.1 fixed
(factory kwargs, hard-coded paths, `.cuda()` labels, per-step prints, logger misuse) and the
reference's flag surface (cet_pick_amd/opts.py).  One process per GPU; under torch.distributed.run
the ranks exchange gradients / keys / SyncBN sums over RCCL.

Data: `--dataset synthetic` trains on a synthetic tomogram; any other `--dataset` value reads the reference's
`<cwd>/data/<--train_img_txt>` list of MRC reconstructions (datasets/tomo_files.py: device loader -> DoG picks -> crop
kernel, same batch contract; the random augmentations of the reference's datasets are out of scope).
"""
import os
import random

import numpy as np
import torch
import torch.distributed as dist

from . import hipops as H
from .datasets.synthetic_moco import SyntheticMocoLoader
from .models.moco import MoCo
from .models.model import create_model, load_model, save_model
from .opts import opts
from .trains.train_factory import train_factory
from .utils.utils import adjust_learning_rate


def build(opt):
    """Everything `main` sets up before its epoch loop (moco_main.py:27-130): process group, the two encoders under MoCo, SGD,
    checkpoint resume, trainer (step engine), loader.  -> (opt, model, optimizer, trainer, loader, start_epoch, rank, world)."""
    torch.manual_seed(opt.seed)
    np.random.seed(opt.seed)
    random.seed(opt.seed)
    opt.distributed = int(os.environ.get("WORLD_SIZE", "1")) > 1
    rank, world = 0, 1
    if opt.distributed:
        opt.gpu = int(os.environ.get("LOCAL_RANK", max(opt.local_rank, 0)))
        torch.cuda.set_device(opt.gpu)
        dist.init_process_group(backend=opt.dist_backend, init_method=opt.dist_url)
        dist.barrier()
        rank, world = dist.get_rank(), dist.get_world_size()
    else:
        opt.gpu = opt.gpus[0] if opt.gpus[0] >= 0 else None
    if opt.gpu is None:
        raise RuntimeError("the MI355X path has no CPU mode (--gpus -1)")
    opt.device = torch.device("cuda", opt.gpu)

    class _DS:                                   # dataset info of task 'moco' (opts.py:321)
        default_resolution, num_classes = [32, 32], 256
    opt = opts().update_dataset_info_and_set_heads(opt, _DS)
    os.makedirs(opt.save_dir, exist_ok=True)

    model_q = create_model(opt.arch, opt.heads, opt.head_conv, last_k=opt.last_k, local_path=opt.pretrained_model)
    model_k = create_model(opt.arch, opt.heads, opt.head_conv, last_k=opt.last_k, local_path=opt.pretrained_model)
    if opt.distributed:
        H.convert_sync_batchnorm(model_q)
        H.convert_sync_batchnorm(model_k)
    model = MoCo(model_q, model_k, dim=128)
    optimizer = torch.optim.SGD(model.parameters(), opt.lr)
    start_epoch = 0
    if opt.load_model != "":
        model, optimizer, start_epoch = load_model(model, opt.load_model, optimizer, opt.resume, opt.lr, opt.lr_step)
    trainer = train_factory[opt.task](opt, model, optimizer)
    if opt.distributed:
        trainer.set_distributed_device(opt.gpu)
        trainer.engine.broadcast_state(0)
    else:
        trainer.set_device(opt.gpus, opt.chunk_sizes, opt.device)
    from .datasets.tomo_files import use_files
    if not use_files(opt):
        loader = SyntheticMocoLoader(batch_size=opt.batch_size, seed=opt.seed, device=opt.device, rank=rank, world=world,
                                     n_crops=max(opt.batch_size * 8, 256))
    else:
        from .datasets.tomo_files import TomoFileMocoLoader
        loader = TomoFileMocoLoader(opt, crop=opt.bbox, device=opt.device, rank=rank, world=world)
    return opt, model, optimizer, trainer, loader, start_epoch, rank, world


def main(opt):
    opt, model, optimizer, trainer, loader, start_epoch, rank, world = build(opt)
    log = open(os.path.join(opt.save_dir, "log.txt"), "a") if rank == 0 else None
    for epoch in range(start_epoch + 1, opt.num_epochs + 1):
        np.random.seed(epoch)
        random.seed(epoch)
        loader.set_epoch(epoch)
        adjust_learning_rate(opt, optimizer, epoch)
        log_dict, _ = trainer.train(epoch, loader)
        if rank == 0:
            line = "epoch: {} |".format(epoch) + "".join("{} {:8f} | ".format(k, v) for k, v in log_dict.items())
            print(line)
            log.write(line + "\n")
            log.flush()
            # moco_main.py:157,175-188: every val_intervals epochs 'model_{last|epoch}.pth' (the file
            # --resume picks up), otherwise 'model_last_contrastive.pth'
            mark = epoch if opt.save_all else "last"
            if opt.val_intervals > 0 and epoch % opt.val_intervals == 0:
                save_model(os.path.join(opt.save_dir, "model_{}.pth".format(mark)), epoch, model, optimizer)
            else:
                save_model(os.path.join(opt.save_dir, "model_last_contrastive.pth"), epoch, model, optimizer)
            if epoch in opt.lr_step:
                save_model(os.path.join(opt.save_dir, "model_{}.pth".format(epoch)), epoch, model, optimizer)
    if log:
        log.close()
    if trainer.engine is not None:
        trainer.engine.close()               # the captured step (and its RCCL work) goes before the communicator
    if opt.distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main(opts().parse())
