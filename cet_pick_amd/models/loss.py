"""Detector-training losses on the MI355X: mirror of the reference's `cet_pick/models/loss.py` for the classes the
semi-supervised trainer uses (trains/tomo_cr_semi_trainer.py:23-41): `_neg_loss` / `FocalLoss` (:378-411),
`_pu_neg_loss` / `PULoss` (:255-324), `ConsistencyLoss` (:701-712), `UnbiasedConLoss` (:571-699).

The voxel losses are single fused reductions (csrc/loss_ops.hip) with hand-written backward kernels; the
data-dependent branch of the PU risk is taken on the device.  `UnbiasedConLoss` never builds the (2N)^2 similarity
matrix: an MFMA kernel streams its tiles and returns the four row sums the loss is a function of; the O(N) tail
(clamps, logs, masked means) is ordinary tensor arithmetic on those vectors.
"""
import numpy as np
import torch
import torch.nn as nn

from .. import _lib as L


def _flat(t, name):
    L.require_cuda(t, name)
    return t.contiguous().view(-1)


class _VoxelLossFn(torch.autograd.Function):
    """mode: 'pu' | 'focal' | 'mse'."""

    @staticmethod
    def forward(ctx, pred, gt, mode, tau, beta):
        p, g = _flat(pred, "pred"), _flat(gt, "gt")
        if p.numel() != g.numel():
            raise ValueError("pred and gt differ in size: %s vs %s" % (tuple(pred.shape), tuple(gt.shape)))
        n = p.numel()
        lib = L.lib()
        ws = L.workspace(lib.mi_voxel_loss_workspace_bytes(n), p.device, "voxloss")
        sums = torch.empty(11, dtype=torch.float64, device=p.device)
        loss = torch.empty((), dtype=torch.float32, device=p.device)
        if mode == "pu":
            rc = lib.mi_pu_focal_loss_fwd(L.ptr(p), L.ptr(g), n, float(tau), float(beta), L.ptr(sums), L.ptr(loss),
                                          L.ptr(ws), ws.numel(), L.stream())
        elif mode == "focal":
            rc = lib.mi_focal_loss_fwd(L.ptr(p), L.ptr(g), n, L.ptr(sums), L.ptr(loss), L.ptr(ws), ws.numel(), L.stream())
        else:
            rc = lib.mi_mse_loss_fwd(L.ptr(p), L.ptr(g), n, L.ptr(sums), L.ptr(loss), L.ptr(ws), ws.numel(), L.stream())
        L.check(rc, "voxel loss fwd")
        ctx.save_for_backward(p, g, sums)
        ctx.mode, ctx.tau, ctx.shape_p, ctx.shape_g = mode, float(tau), pred.shape, gt.shape
        ctx.g_needs = gt.requires_grad
        ctx.sums = sums
        return loss

    @staticmethod
    def backward(ctx, dloss):
        p, g, sums = ctx.saved_tensors
        n = p.numel()
        lib = L.lib()
        dl = dloss.contiguous().float()
        dp = torch.empty_like(p)
        dg = None
        if ctx.mode == "pu":
            rc = lib.mi_pu_focal_loss_bwd(L.ptr(p), L.ptr(g), n, ctx.tau, L.ptr(sums), L.ptr(dl), L.ptr(dp), L.stream())
        elif ctx.mode == "focal":
            rc = lib.mi_focal_loss_bwd(L.ptr(p), L.ptr(g), n, L.ptr(sums), L.ptr(dl), L.ptr(dp), L.stream())
        else:
            dg = torch.empty_like(g) if ctx.g_needs else None
            rc = lib.mi_mse_loss_bwd(L.ptr(p), L.ptr(g), n, L.ptr(sums), L.ptr(dl), L.ptr(dp), L.ptr(dg), L.stream())
        L.check(rc, "voxel loss bwd")
        return dp.view(ctx.shape_p), (dg.view(ctx.shape_g) if dg is not None else None), None, None, None


_NO_POS = ("Num of true positive is zero, please check tomogram size of input coordinates order or use smaller "
           "translation ratio")


def _neg_loss(pred, gt):
    """loss.py:378-411."""
    return _VoxelLossFn.apply(pred, gt, "focal", 0.0, 0.0)


def _pu_neg_loss(pred, gt, tau, beta, gamma, check_positives=True):
    """loss.py:255-308.  check_positives: the reference's ValueError on zero positives (one host sync)."""
    if check_positives and not bool((gt == 1).any()):
        raise ValueError(_NO_POS)
    return _VoxelLossFn.apply(pred, gt, "pu", tau, beta)


class FocalLoss(nn.Module):
    def forward(self, out, target):
        return _neg_loss(out, target)


class PULoss(nn.Module):
    def __init__(self, tau, beta=0, gamma=1):
        super().__init__()
        self.tau, self.beta, self.gamma = tau, beta, gamma
        self.puloss = _pu_neg_loss

    def forward(self, pred, gt):
        return self.puloss(pred, gt, self.tau, self.beta, self.gamma)


class ConsistencyLoss(nn.Module):
    def forward(self, out_prob, out_prob_cr):
        return _VoxelLossFn.apply(out_prob, out_prob_cr, "mse", 0.0, 0.0)


class _UclRowSumsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat, cls, inv_T):
        n2, dim = feat.shape
        dev = feat.device
        outs = [torch.empty(n2, dtype=torch.float32, device=dev) for _ in range(5)]
        L.check(L.lib().mi_ucl_rowsums_fwd(L.ptr(feat), L.ptr(cls), n2, dim, float(inv_T), *[L.ptr(o) for o in outs],
                                           L.stream()), "mi_ucl_rowsums_fwd")
        ctx.save_for_backward(feat, cls, outs[0])
        ctx.inv_T = float(inv_T)
        ctx.mark_non_differentiable(outs[0])
        return tuple(outs)

    @staticmethod
    def backward(ctx, g_max, g_all, g_pos, g_other, g_pair):
        feat, cls, rowmax = ctx.saved_tensors
        n2, dim = feat.shape
        z = lambda g: torch.zeros(n2, dtype=torch.float32, device=feat.device) if g is None else g.contiguous().float()
        g_all, g_pos, g_other, g_pair = z(g_all), z(g_pos), z(g_other), z(g_pair)
        dfeat = torch.empty_like(feat)
        rng = torch.empty(2, dtype=torch.float32, device=feat.device)      # (largest row maximum, fast-path flag): decided on the device
        L.check(L.lib().mi_ucl_rowsums_bwd_ranged(L.ptr(feat), L.ptr(cls), n2, dim, ctx.inv_T, L.ptr(rowmax), L.ptr(g_all),
                                                  L.ptr(g_pos), L.ptr(g_other), L.ptr(g_pair), L.ptr(dfeat), L.ptr(rng), L.stream()),
                "mi_ucl_rowsums_bwd_ranged")
        return dfeat, None, None


class UnbiasedConLoss(nn.Module):
    """loss.py:571-699: debiased contrastive regularisation; returns (supervised, unsupervised) terms."""

    def __init__(self, base_temperature, class_prob):
        super().__init__()
        self.base_temperature = base_temperature
        self.tau_plus = class_prob

    def calc_g(self, pos_mean, neg_mean, class_prob):
        Ng = (neg_mean - class_prob * pos_mean) / (1 - class_prob)
        return torch.clamp(Ng, min=np.e ** (-1 / self.base_temperature))

    def forward(self, labels, out_labels, out_labels_cr, all_features, all_features_cr, opt):
        labels = labels.reshape(-1)
        n = all_features.shape[0]
        pos1 = labels.gt(opt.thresh) if opt.thresh < 1 else labels.eq(1)
        num_of_positives = pos1.sum()
        if num_of_positives == 0:            # the reference's host-side check (loss.py:607-608)
            raise ValueError(_NO_POS)
        num_of_negatives = 2 * (n - num_of_positives)
        feat = torch.cat([all_features, all_features_cr], dim=0).float()
        dim = feat.shape[1]
        if dim not in (32, 64):              # kernel widths; zero columns do not change any inner product
            pad = (32 if dim < 32 else 64) - dim
            if pad < 0:
                raise L.HipExtensionError("feature dim %d > 64 is not supported by mi_ucl_rowsums" % dim)
            feat = torch.nn.functional.pad(feat, (0, pad))
        feat = L.require_cuda(feat, "features").contiguous()
        all_labels = torch.cat([labels, labels], dim=0)
        all_out_preds = torch.cat([out_labels.reshape(-1), out_labels_cr.reshape(-1)], dim=0)
        pos = all_labels.gt(opt.thresh) if opt.thresh < 1 else all_labels.eq(1)
        other = all_labels.lt(opt.thresh)
        un = all_labels.lt(0)
        cls = (pos.to(torch.uint8) | (other.to(torch.uint8) << 1)).contiguous()
        _, s_all, s_pos, s_other, e_pair = _UclRowSumsFn.apply(feat, cls, 1.0 / self.base_temperature)

        posf, unf = pos.float(), un.float()
        # supervised term over the positive anchors (loss.py:644-650)
        pos_feat_mean = s_pos / (posf.sum() - 1)
        rem_feat_mean = s_other / other.float().sum()
        Ng = self.calc_g(pos_feat_mean, rem_feat_mean, self.tau_plus)
        sup_rows = -torch.log(pos_feat_mean / (pos_feat_mean + Ng))
        debiased_loss_sup = (torch.where(pos, sup_rows, torch.zeros_like(sup_rows))).sum() / posf.sum()

        # unsupervised term over the unlabeled anchors (loss.py:653-694); masked means instead of boolean indexing
        up = e_pair
        urem = (s_all - e_pair) / num_of_negatives
        Ng_pos = self.calc_g(up, urem, self.tau_plus)
        Ng_neg = self.calc_g(up, urem, 1 - self.tau_plus)
        lpos = -torch.log(up / (up + Ng_pos)) * all_out_preds
        lneg = -torch.log(up / (up + Ng_neg)) * (1 - all_out_preds)

        def masked_mean(v, m):
            cnt = m.float().sum()
            s = torch.where(m, v, torch.zeros_like(v)).sum()
            return torch.where(cnt > 0, s / cnt.clamp(min=1), torch.zeros_like(s))

        m_hi = un & all_out_preds.gt(0.99)
        m_lo = un & all_out_preds.lt(0.01)
        m_mid = un & all_out_preds.gt(0.01) & all_out_preds.lt(0.99)
        debiased_loss_unsup = masked_mean(lpos, m_hi) + masked_mean(lneg, m_lo) + masked_mean(lpos, m_mid) + \
            masked_mean(lneg, m_mid)
        return debiased_loss_sup, debiased_loss_unsup
