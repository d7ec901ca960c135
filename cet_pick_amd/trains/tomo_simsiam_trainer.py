"""Mirror of cet_pick/trains/tomo_simsiam_trainer.py (reference :17-55): SimSiam cosine loss
-(mean cos(p1,z2) + mean cos(p2,z1))/2 and the `output_std` collapse monitor."""
import torch

from .. import hipops as H
from .base_trainer import BaseTrainer


def _neg_mean_cosine(p, z):
    """-mean_b cos(p_b, z_b); z carries no gradient (it is the detached projection)."""
    pn = H.l2_normalize(p.contiguous())
    zn = H.l2_normalize(z.detach().contiguous())
    return -H.rowdot_mean(pn, zn)


class TomoSimSiamLoss(torch.nn.Module):
    def __init__(self, opt):
        super().__init__()
        self.opt = opt

    def forward(self, outputs, batch, epoch):
        p1, z1 = outputs[0]["pred"], outputs[0]["proj"]
        p2, z2 = outputs[1]["pred"], outputs[1]["proj"]
        cos_loss = (_neg_mean_cosine(p1, z2) + _neg_mean_cosine(p2, z1)) * 0.5
        with torch.no_grad():
            output_std = H.column_std_mean(H.l2_normalize(p1.detach().contiguous()))
        loss_stats = {"loss": cos_loss, "cosine_loss": cos_loss, "output_std": output_std}
        return cos_loss, loss_stats


class TomoSimSiamTrainer(BaseTrainer):
    def __init__(self, opt, model, optimizer=None):
        super().__init__(opt, model, optimizer=optimizer)

    def _get_losses(self, opt):
        return ["loss", "cosine_loss", "output_std"], TomoSimSiamLoss(opt)

    def debug(self, batch, output, iter_id):
        pass

    def save_results(self, output, batch, results):
        pass
