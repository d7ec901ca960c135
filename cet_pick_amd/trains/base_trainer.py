"""Mirror of the hot loop of cet_pick/trains/base_trainer.py (reference :124-133 ModelWithLossSimSiam,
:205-249 BaseTrainer.__init__/set_device/set_distributed_device, :446-552 run_epoch, :563-572 val/train).

Same call surface: `Trainer(opt, model, optimizer)`, `.set_device(gpus, chunk_sizes, device)`,
`.set_distributed_device(gpu)`, `.train(epoch, loader) -> (dict(loss_stats..., 'time'), results)`.
Differences that make it an MI355X path rather than a copy:
  * a MoCo model runs through `MocoStepEngine` (flat arenas, fused SGD, optional hipGraph);
  * data parallelism is the engine's explicit RCCL exchange (gradient arena all-reduce, key
    all-gather, SyncBN sums) instead of a DistributedDataParallel wrapper;
  * loss statistics stay on the device and are read back once per `print_iter` / epoch, not with
    three `.item()` syncs per iteration (base_trainer.py:514-530).
"""
import time

import torch

from .. import hipops as H
from ..models.moco import MoCo
from ..utils.utils import AverageMeter
from .moco_engine import MocoStepEngine


class ModelWithLossSimSiam(torch.nn.Module):
    """base_trainer.py:124-133."""

    def __init__(self, model, loss):
        super().__init__()
        self.model = model
        self.loss = loss

    def forward(self, batch, epoch, phase):
        outputs = self.model(batch["input"], batch["input_aug"])
        loss, loss_stats = self.loss(outputs, batch, epoch)
        return outputs, loss, loss_stats


class ModelWithLossSimSiam2D3D(torch.nn.Module):
    """base_trainer.py:113-122: tilt-series and tomogram patches of both views."""

    def __init__(self, model, loss):
        super().__init__()
        self.model = model
        self.loss = loss

    def forward(self, batch, epoch, phase):
        outputs = self.model(batch["input"], batch["input_3d"], batch["input_aug"], batch["input_aug_3d"])
        loss, loss_stats = self.loss(outputs, batch, epoch)
        return outputs, loss, loss_stats


class ModelWithLoss(torch.nn.Module):
    """base_trainer.py:134-154: the detector tasks - two forward passes (input and its augmented view) in training,
    one without gradients (and in eval mode) otherwise."""

    def __init__(self, model, loss):
        super().__init__()
        self.model = model
        self.loss = loss

    def forward(self, batch, epoch, phase):
        if phase == "train":
            outputs = self.model(batch["input"])
            outputs_cr = self.model(batch["input_aug"])
            loss, loss_stats = self.loss(outputs, batch, epoch, phase, output_cr=outputs_cr)
        else:
            with torch.no_grad():
                self.model.eval()
                outputs = self.model(batch["input"])
                loss, loss_stats = self.loss(outputs, batch, epoch, phase, output_cr=None)
        return outputs[-1], loss, loss_stats


class BaseTrainer(object):
    def __init__(self, opt, model, optimizer=None):
        self.opt = opt
        self.optimizer = optimizer
        self.loss_stats, self.loss = self._get_losses(opt)
        self.iter = 0
        if opt.task in ("simsiam", "moco", "simsiam3d"):
            self.model_with_loss = ModelWithLossSimSiam(model, self.loss)
        elif opt.task == "simsiam2d3d":
            self.model_with_loss = ModelWithLossSimSiam2D3D(model, self.loss)
        elif opt.task in ("semi", "tomo", "semi3d"):
            self.model_with_loss = ModelWithLoss(model, self.loss)
        else:
            raise NotImplementedError("task '%s' is outside the hot path built here (DESIGN.md §7)" % opt.task)
        self.engine = None
        self.exchange = None        # hipops.GradExchange of a non-MoCo model under torch.distributed
        self.device = None

    # ---- device placement ----------------------------------------------------------------------
    def _lr(self):
        if self.optimizer is not None:
            return self.optimizer.param_groups[0]["lr"]
        return self.opt.lr

    def _make_engine(self):
        model = self.model_with_loss.model
        if isinstance(model, MoCo):
            wd = self.optimizer.param_groups[0].get("weight_decay", 0.0) if self.optimizer is not None else 0.0
            self.engine = MocoStepEngine(model, lr=self._lr(), weight_decay=wd,
                                         use_graph=bool(getattr(self.opt, "hipgraph", False)))
        elif self._simsiam_engine_ok(model):
            # two-view SimSiam models under plain SGD: flat arenas, fused optimizer, device meters, hipGraph (trains/simsiam_engine.py)
            from .simsiam_engine import SimSiamStepEngine
            wd = self.optimizer.param_groups[0].get("weight_decay", 0.0)
            self.engine = SimSiamStepEngine(self.model_with_loss, lr=self._lr(), weight_decay=wd,
                                            use_graph=bool(getattr(self.opt, "hipgraph", False)))
            if H._distributed():
                self.engine.broadcast_state(0)
        elif H._distributed():
            # every other task: stock optimizer, gradients averaged over the ranks between backward() and step()
            self.exchange = H.GradExchange(model)
            self.exchange.broadcast_parameters(0)
            import torch.distributed as dist
            for b in model.buffers():
                dist.broadcast(b, 0)

    def _simsiam_engine_ok(self, model):
        """The step engine replaces `zero_grad / backward / optimizer.step` only where it is the same arithmetic: a two-view SimSiam
        model on the GPU under torch.optim.SGD without momentum / dampening / nesterov, every parameter trained by that optimizer."""
        import os
        if os.environ.get("CETPICK_SIMSIAM_ENGINE", "1") == "0" or self.opt.task not in ("simsiam", "simsiam3d", "simsiam2d3d"):
            return False
        o = self.optimizer
        if not isinstance(o, torch.optim.SGD) or len(o.param_groups) != 1:
            return False
        g = o.param_groups[0]
        if g.get("momentum", 0) or g.get("dampening", 0) or g.get("nesterov", False) or g.get("maximize", False):
            return False
        params = list(model.parameters())
        return (len(params) > 0 and all(p.is_cuda and p.requires_grad and p.dtype == torch.float32 for p in params)
                and len(g["params"]) == len(params))

    def close(self):
        """Release what refers to the process group before it is destroyed (a captured step holds RCCL work)."""
        if self.engine is not None:
            self.engine.close()

    def set_device(self, gpus, chunk_sizes, device):
        """base_trainer.py:240-249 (single process).  More than one GPU per process (the reference's
        nn.DataParallel branch) is replaced by one process per GPU: use torch.distributed."""
        if len(gpus) > 1:
            raise NotImplementedError("one process per GPU: launch with torch.distributed.run")
        self.device = torch.device(device)
        self.model_with_loss = self.model_with_loss.to(self.device)
        self._make_engine()

    def set_distributed_device(self, gpus):
        """base_trainer.py:229-238: one rank per GPU.  SyncBN is already selected by
        hipops.convert_sync_batchnorm (moco_main.py:65-66); gradients are exchanged by the engine."""
        if gpus is not None:
            torch.cuda.set_device(gpus)
            self.device = torch.device("cuda", gpus)
        else:
            self.device = torch.device("cuda")
        self.model_with_loss.to(self.device)
        self._make_engine()

    # ---- the hot loop ----------------------------------------------------------------------------
    def train_step(self, x1, x2, eager=False):
        """One training step on a resident pair of view batches, as run_epoch's loop body performs it (the bench's step loop and
        the engine-equals-plain-sequence tests).  `eager`: keep a step engine out of its hipGraph for this call.  -> loss (device)."""
        if self.engine is not None:
            if eager and hasattr(self.engine, "step_eager"):
                return self.engine.step_eager(x1, x2)
            return self.engine.step(x1, x2)
        self.model_with_loss.train()
        _, loss, _ = self.model_with_loss({"input": x1, "input_aug": x2}, 0, "train")
        self.optimizer.zero_grad()
        loss.backward()
        if self.exchange is not None:
            self.exchange.sync()
        self.optimizer.step()
        return loss.detach()

    def run_epoch(self, phase, epoch, data_loader):
        opt = self.opt
        mwl = self.model_with_loss
        mwl.train() if phase == "train" else mwl.eval()
        results = {}
        data_time, batch_time = AverageMeter(), AverageMeter()
        avg_loss_stats = {l: AverageMeter() for l in self.loss_stats}
        num_iters = len(data_loader) if opt.num_iters < 0 else opt.num_iters
        dev_sums = {l: None for l in self.loss_stats}       # device-side accumulation (no per-iter sync)
        n_pending = 0
        t0 = end = time.time()

        engine_sums = phase == "train" and self.engine is not None and hasattr(self.engine, "take_loss_sum")
        engine_stats = phase == "train" and self.engine is not None and hasattr(self.engine, "take_stat_sums")
        if engine_sums:
            self.engine.take_loss_sum()                     # (steps taken outside this epoch)
        if engine_stats:
            self.engine.take_stat_sums()

        def flush():
            nonlocal n_pending
            if n_pending:
                if engine_sums:                             # the step engine accumulates its loss inside the step (one graph node)
                    assert list(self.loss_stats) == ["loss"] or set(self.loss_stats) <= {"loss", "infoNCE"}, self.loss_stats
                    tot = self.engine.take_loss_sum()
                    for l in self.loss_stats:
                        avg_loss_stats[l].update(tot / n_pending, n_pending)
                elif engine_stats:                          # ... and the SimSiam engine one sum per loss statistic
                    tot = self.engine.take_stat_sums()
                    for l in self.loss_stats:
                        avg_loss_stats[l].update(tot[l] / n_pending, n_pending)
                else:
                    for l in self.loss_stats:
                        avg_loss_stats[l].update(float(dev_sums[l].item()) / n_pending, n_pending)
                        dev_sums[l] = None
                n_pending = 0

        if self.engine is not None and self.optimizer is not None:
            self.engine.set_lr(self._lr())                  # adjust_learning_rate acted on the optimizer
        for iter_id, batch in enumerate(data_loader):
            if iter_id >= num_iters:
                break
            data_time.update(time.time() - end)
            for k in batch:
                if k != "meta" and isinstance(batch[k], torch.Tensor):
                    batch[k] = batch[k].to(device=self.device, non_blocking=True)
            if phase == "train" and self.engine is not None:
                if hasattr(self.engine, "step_batch"):
                    loss = self.engine.step_batch(batch)
                else:
                    loss = self.engine.step(batch["input"], batch["input_aug"])
                loss_stats = {l: loss for l in self.loss_stats}
            else:
                with torch.set_grad_enabled(phase == "train"):
                    output, loss, loss_stats = mwl(batch, epoch, phase)
                if phase == "train":
                    self.optimizer.zero_grad()
                    loss.backward()
                    if self.exchange is not None:
                        self.exchange.sync()
                    self.optimizer.step()
            if not engine_sums and not engine_stats:
                for l in self.loss_stats:
                    v = loss_stats[l].detach().float().mean()
                    dev_sums[l] = v.clone() if dev_sums[l] is None else dev_sums[l] + v
            n_pending += 1
            batch_time.update(time.time() - end)
            end = time.time()
            if opt.print_iter > 0 and iter_id % opt.print_iter == 0:
                flush()
                msg = "{}/{}| {}: [{}][{}/{}]".format(opt.task, opt.exp_id, phase, epoch, iter_id, num_iters)
                for l in avg_loss_stats:
                    msg += "|{} {:.4f} ".format(l, avg_loss_stats[l].avg)
                if not opt.hide_data_time:
                    msg += "|Data {dt.val:.3f}s({dt.avg:.3f}s) |Net {bt.avg:.3f}s".format(dt=data_time, bt=batch_time)
                print(msg)
        flush()
        ret = {k: v.avg for k, v in avg_loss_stats.items()}
        ret["time"] = (time.time() - t0) / 60.0
        return ret, results

    def debug(self, batch, output, iter_id):
        raise NotImplementedError

    def save_result(self, output, batch, results):
        raise NotImplementedError

    def _get_losses(self, opt):
        raise NotImplementedError

    def val(self, epoch, data_loader):
        return self.run_epoch("val", epoch, data_loader)

    def train(self, epoch, data_loader):
        return self.run_epoch("train", epoch, data_loader)
