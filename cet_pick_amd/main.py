"""`python -m cet_pick_amd.main semi --arch unet_4 --contrastive ...` - the reference's cet_pick/main.py (:21-129):
CenterNet-3D detector training (task 'semi': PU focal heat-map loss + debiased contrastive regularisation +
consistency) with Adam, validation every `--val_intervals` epochs, the reference's checkpoint names
(`model_last.pth`, `model_best_contrastive.pth`, `model_last_contrastive.pth`, `model_<epoch>.pth` at the lr steps), the 10x lr
drops of `--lr_step` and the `log.txt` line.  One process per GPU; under torch.distributed.run (BASELINE config 5) the
ranks draw different crop pairs and exchange gradients over RCCL (hipops.GradExchange) with SyncBN statistics
(main.py:34-56).

The reference datasets are out of scope: labelled synthetic tomograms (datasets/synthetic_datasets.py), same batch contract.
"""
import os

import torch
import torch.distributed as dist

from . import hipops as H
from .datasets.synthetic_datasets import SyntheticDetectorDataset
from .models.model import create_model, load_model, save_model
from .opts import opts
from .simsiam_main import init_distributed
from .trains.train_factory import train_factory
from .utils.utils import TextLog


def main(opt):
    torch.manual_seed(opt.seed)
    rank, world = init_distributed(opt)
    Dataset = SyntheticDetectorDataset
    opt = opts().update_dataset_info_and_set_heads(opt, Dataset)
    logger = TextLog(opt, enabled=rank == 0)

    print("Creating model...")
    model = create_model(opt.arch, opt.heads, opt.head_conv, last_k=opt.last_k)
    if opt.distributed:
        H.convert_sync_batchnorm(model)
    optimizer = torch.optim.Adam(filter(lambda p: p.requires_grad, model.parameters()), opt.lr)
    start_epoch = 0
    if opt.load_model != "":
        model, optimizer, start_epoch = load_model(model, opt.load_model, optimizer, opt.resume, opt.lr, opt.lr_step)

    trainer = train_factory[opt.task](opt, model, optimizer)
    if opt.distributed:
        trainer.set_distributed_device(opt.gpu)
    else:
        trainer.set_device(opt.gpus, opt.chunk_sizes, opt.device)

    print("Setting up data...")
    val_set = Dataset(opt, "train", device=opt.device, per_epoch=4 * max(1, opt.batch_size), rank=0, world=1)
    val_set.set_epoch(10 ** 6)                    # a fixed held-out draw of crop pairs
    if opt.test:
        log_dict_val, _ = trainer.val(0, val_set)
        print(log_dict_val)
        return
    train_set = Dataset(opt, "train", device=opt.device, rank=rank, world=world)

    print("Starting training...")
    best = 1e10
    for epoch in range(start_epoch + 1, opt.num_epochs + 1):
        mark = epoch if opt.save_all else "last"
        train_set.set_epoch(epoch)
        log_dict_train, _ = trainer.train(epoch, train_set)
        logger.write("epoch: {} |".format(epoch))
        for k, v in log_dict_train.items():
            logger.write("{} {:8f} | ".format(k, v))
        if opt.val_intervals > 0 and epoch % opt.val_intervals == 0:
            if rank == 0:
                save_model(os.path.join(opt.save_dir, "model_{}.pth".format(mark)), epoch, model, optimizer)
            with torch.no_grad():
                log_dict_val, _ = trainer.val(epoch, val_set)
            for k, v in log_dict_val.items():
                logger.write("{} {:8f} | ".format(k, v))
            if log_dict_val[opt.metric] < best:
                best = log_dict_val[opt.metric]
                if rank == 0:
                    save_model(os.path.join(opt.save_dir, "model_best_contrastive.pth"), epoch, model)
        elif rank == 0:
            save_model(os.path.join(opt.save_dir, "model_last_contrastive.pth"), epoch, model, optimizer)
        logger.write("\n")
        if epoch in opt.lr_step:
            if rank == 0:
                save_model(os.path.join(opt.save_dir, "model_{}.pth".format(epoch)), epoch, model, optimizer)
            lr = opt.lr * (0.1 ** (opt.lr_step.index(epoch) + 1))
            print("Drop LR to", lr)
            for param_group in optimizer.param_groups:
                param_group["lr"] = lr
    logger.close()
    if opt.distributed:
        trainer.close()
        dist.destroy_process_group()


if __name__ == "__main__":
    main(opts().parse())
