"""Mirror of the hot-path helpers of cet_pick/utils/utils.py."""
import math

import numpy as np


class AverageMeter(object):
    """Running average (reference utils/utils.py AverageMeter, used by run_epoch)."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        if self.count > 0:
            self.avg = self.sum / self.count


def learning_rate_at(args, epoch):
    """utils/utils.py:58-67: cosine to lr*decay^3, or step decay by lr_decay_rate per passed lr_step."""
    lr = args.lr
    if args.cosine:
        eta_min = lr * (args.lr_decay_rate ** 3)
        lr = eta_min + (lr - eta_min) * (1 + math.cos(math.pi * epoch / args.num_epochs)) / 2
    else:
        steps = np.sum(epoch > np.asarray(args.lr_step))
        if steps > 0:
            lr = lr * (args.lr_decay_rate ** steps)
    return float(lr)


def adjust_learning_rate(args, optimizer, epoch):
    """utils/utils.py:58-70.  `optimizer` is a torch optimizer (param_groups) or a step engine with
    `set_lr` (cet_pick_amd.trains.moco_engine.MocoStepEngine)."""
    lr = learning_rate_at(args, epoch)
    if hasattr(optimizer, "set_lr"):
        optimizer.set_lr(lr)
    else:
        for param_group in optimizer.param_groups:
            param_group["lr"] = lr
    return lr


class TextLog(object):
    """The `log.txt` of the reference's Logger (logger.py: `write` appends to save_dir/log.txt); tensorboard summaries are
    out of scope, `scalar_summary` is accepted and ignored."""

    def __init__(self, opt, enabled=True):
        import os
        self.f = None
        if enabled:
            os.makedirs(opt.save_dir, exist_ok=True)
            self.f = open(os.path.join(opt.save_dir, "log.txt"), "a")

    def write(self, txt):
        if self.f:
            self.f.write(txt)
            if txt.endswith("\n"):
                self.f.flush()

    def scalar_summary(self, tag, value, step):
        pass

    def close(self):
        if self.f:
            self.f.close()
            self.f = None
