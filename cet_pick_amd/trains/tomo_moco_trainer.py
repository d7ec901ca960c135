"""Mirror of cet_pick/trains/tomo_moco_trainer.py (reference :17-93)."""
import torch

from .. import hipops as H
from .base_trainer import BaseTrainer


class TomoMocoLoss(torch.nn.Module):
    """:17-77: CrossEntropyLoss(logits, labels) -> (loss, {'loss', 'infoNCE'}).  MoCo's labels are
    all zero (models/moco.py:141), which is what the fused kernel assumes.  There is no host path: a
    logits tensor that is not on the GPU raises (hipops.HipExtensionError)."""

    def __init__(self, opt):
        super().__init__()
        self.opt = opt

    def forward(self, outputs, batch, epoch):
        logits, labels = outputs[0], outputs[1]
        loss = H.cross_entropy_label0(logits.contiguous())
        return loss, {"loss": loss, "infoNCE": loss}


class TomoMocoTrainer(BaseTrainer):
    def __init__(self, opt, model, optimizer=None):
        super().__init__(opt, model, optimizer=optimizer)

    def _get_losses(self, opt):
        return ["loss", "infoNCE"], TomoMocoLoss(opt)

    def debug(self, batch, output, iter_id):
        pass

    def save_results(self, output, batch, results):
        pass
