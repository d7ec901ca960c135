"""Mirror of the loss half of `cet_pick/trains/tomo_cr_semi_trainer.py` (`TomoCRSemiLoss` :18-112): PU focal risk on
the heat-map + debiased contrastive regularisation between the two augmented views + consistency, composed from the
HIP losses in models/loss.py.  The `--pn` (SupConLossV2_more) and `--ge` (PUGELoss) variants are outside the hot path.

`TomoCRSemiTrainer` drives it through the shared `BaseTrainer` loop (two forward passes per step).
"""
import torch

from .base_trainer import BaseTrainer
from ..models.loss import ConsistencyLoss, FocalLoss, PULoss, UnbiasedConLoss
from ..models.utils import _sigmoid


class TomoCRSemiLoss(torch.nn.Module):
    def __init__(self, opt):
        super().__init__()
        if opt.pn or opt.ge:
            raise NotImplementedError("--pn / --ge loss variants of the reference are outside the MI355X hot path")
        self.crit = PULoss(opt.tau)
        self.crit2 = FocalLoss()
        self.cr_loss = UnbiasedConLoss(opt.temp, opt.tau)
        self.cons_loss = ConsistencyLoss()
        self.opt = opt

    def forward(self, outputs, batch, epoch, phase, output_cr=None):
        opt = self.opt
        cr_loss, hm_loss, consis_loss = 0, 0, 0
        for s in range(opt.num_stacks):
            output = outputs[s]
            if output_cr is not None:
                output_cr = output_cr[s]
            output["hm"] = _sigmoid(output["hm"])
            if output_cr is not None:
                output_cr["hm"] = _sigmoid(output_cr["hm"])
        crit = self.crit if phase == "train" else self.crit2
        hm_loss = hm_loss + crit(output["hm"], batch["hm"]) / opt.num_stacks
        if opt.contrastive and phase == "train":
            fm = output["proj"]
            b, ch = fm.shape[:2]
            flip_dim = -2 if batch["flip_prob"] > 0.5 else -1          # undo the second view's flip (:69-74)
            fm_cr = output_cr["proj"].flip(flip_dim)
            hm_cr = output_cr["hm"].flip(flip_dim)
            # (b, ch, d, h, w) -> (b*d*h*w, ch), voxels of a batch element contiguous (:75-80)
            feat = fm.reshape(b, ch, -1).permute(1, 0, 2).reshape(ch, -1).T
            feat_cr = fm_cr.reshape(b, ch, -1).permute(1, 0, 2).reshape(ch, -1).T
            gt_f = batch["hm"].reshape(-1).contiguous()
            hm_f = output["hm"].reshape(-1).contiguous()
            hm_cr_f = hm_cr.reshape(-1).contiguous()
            sup, unsup = self.cr_loss(gt_f, hm_f, hm_cr_f, feat, feat_cr, opt)
            cr_loss = cr_loss + sup + 0.1 * unsup
            consis_loss = consis_loss + self.cons_loss(hm_f, hm_cr_f)
            loss = hm_loss + cr_loss * opt.cr_weight + consis_loss
        else:
            cr_loss = hm_loss * 0
            loss = hm_loss
            consis_loss = hm_loss * 0
        return loss, {"loss": loss, "hm_loss": hm_loss, "cr_loss": cr_loss, "consis_loss": consis_loss}


class TomoCRSemiTrainer(BaseTrainer):
    """tomo_cr_semi_trainer.py:114-125: the semi-supervised detector trainer (task 'semi')."""

    def __init__(self, opt, model, optimizer=None):
        super().__init__(opt, model, optimizer=optimizer)

    def _get_losses(self, opt):
        return ["loss", "hm_loss", "cr_loss", "consis_loss"], TomoCRSemiLoss(opt)
