"""CPU oracle (TEST INFRASTRUCTURE ONLY - see oracle/__init__.py) for the detector-training losses, SURVEY.md §8 row a23.

Plain torch restatement of the reference's cet_pick/models/loss.py: `_neg_loss` :378-411, `_pu_neg_loss` :255-308,
`ConsistencyLoss` :701-712, `UnbiasedConLoss.forward` :594-699 (dense (2N)^2 matrix, fine at test sizes).
Pinned by tests/golden/losses.npz, produced by the reference's own functions (tests/golden/gen_golden.py: gen_losses).
"""
import numpy as np
import torch


def neg_loss(pred, gt):
    gt = gt.unsqueeze(0)
    pos = gt.eq(1).float()
    soft = ((gt.gt(-1).float()) == (gt.lt(1).float())).float()
    pos_loss = (torch.log(pred) * torch.pow(1 - pred, 2) * pos).sum()
    neg_loss_ = (torch.log(1 - pred) * torch.pow(pred, 2) * torch.pow(1 - gt, 4) * soft).sum()
    n = pos.sum()
    return -neg_loss_ if n == 0 else -(pos_loss + neg_loss_) / n


def pu_neg_loss(pred, gt, tau, beta=0.0):
    pred, gt = pred.squeeze(), gt.squeeze()
    pos = gt.eq(1).float()
    soft = ((gt.gt(-1).float()) == (gt.lt(1).float())).float()
    unl = gt.eq(-1).float()
    n_pos, n_soft, n_unl = pos.sum(), soft.sum(), unl.sum()
    if n_pos == 0:
        raise ValueError("no positives")
    a = torch.log(pred) * torch.pow(1 - pred, 2)
    b = torch.log(1 - pred) * torch.pow(pred, 2)
    pos_tot = -(a * pos).sum() / n_pos
    negpos_tot = -(b * pos).sum() / n_pos
    if n_soft > 0:
        pos_tot = pos_tot - (b * torch.pow(1 - gt, 4) * soft).sum() / n_soft
        negpos_tot = negpos_tot - (a * torch.pow(gt, 4) * soft).sum() / n_soft
    pos_risk = pos_tot * tau
    neg_total = -tau * negpos_tot + (-(b * unl).sum()) / n_unl
    return pos_risk if neg_total < -beta else pos_risk + neg_total


def mse(a, b):
    return ((a - b) ** 2).mean()


def unbiased_con_loss(labels, out_labels, out_labels_cr, f, f_cr, T, tau_plus, thresh):
    n = f.shape[0]
    pos1 = labels.gt(thresh) if thresh < 1 else labels.eq(1)
    n_pos1 = pos1.float().sum()
    n_neg = 2 * (n - n_pos1)
    self_mask = torch.zeros(2 * n, 2 * n)
    self_mask[:n, n:] = torch.eye(n)
    self_mask[n:, :n] = torch.eye(n)
    tot = torch.cat([f, f_cr], 0)
    sims = tot @ tot.t() / T
    sims = sims - sims.max(dim=1, keepdim=True)[0].detach()
    sims = torch.exp(sims * (1 - torch.eye(2 * n)))
    all_labels = torch.cat([labels, labels], 0)
    preds = torch.cat([out_labels, out_labels_cr], 0)
    pos = all_labels.gt(thresh) if thresh < 1 else all_labels.eq(1)
    un = all_labels.lt(0)
    other = all_labels.lt(thresh).float()

    def calc_g(p, q, c):
        return torch.clamp((q - c * p) / (1 - c), min=np.e ** (-1 / T))

    pf = sims[pos]
    pos_mean = (pf * pos.float()).sum(1) / (pos.float().sum() - 1)
    rem_mean = (pf * other).sum(1) / other.sum()
    sup = (-torch.log(pos_mean / (pos_mean + calc_g(pos_mean, rem_mean, tau_plus)))).mean()
    uf, um = sims[un], self_mask[un]
    up = (uf * um).sum(1)
    urem = (uf * (1 - um)).sum(1) / n_neg
    gp, gn = calc_g(up, urem, tau_plus), calc_g(up, urem, 1 - tau_plus)
    pr = preds[un]
    lpos = -torch.log(up / (up + gp)) * pr
    lneg = -torch.log(up / (up + gn)) * (1 - pr)
    unsup = torch.zeros(())
    hi, lo = pr.gt(0.99), pr.lt(0.01)
    mid = pr.gt(0.01) & pr.lt(0.99)
    if hi.any():
        unsup = unsup + lpos[hi].mean()
    if lo.any():
        unsup = unsup + lneg[lo].mean()
    if mid.any():
        unsup = unsup + lpos[mid].mean() + lneg[mid].mean()
    return sup, unsup


def unbiased_con_loss_streamed(labels, out_labels, out_labels_cr, f, f_cr, T, tau_plus, thresh, block=2048, device=None,
                               dtype=None):
    """`unbiased_con_loss` (models/loss.py:571-699) without the (2N)^2 matrix in memory: the same rows, `block` of them at
    a time, reduced to the row sums the loss is a function of.  Forward values only (no autograd graph is kept).  It exists
    so that the batch-16 step of BASELINE config C5 (2N = 196,608: a 154 GB matrix) has an oracle at all;
    tests/test_oracle_losses.py pins it to the dense form above on sizes where both run."""
    device = device or f.device
    dtype = dtype or f.dtype
    n = f.shape[0]
    to = lambda t: t.detach().to(device=device, dtype=dtype)
    labels, out_labels, out_labels_cr, f, f_cr = (to(t) for t in (labels, out_labels, out_labels_cr, f, f_cr))
    tot = torch.cat([f, f_cr], 0)
    all_labels = torch.cat([labels, labels], 0)
    preds = torch.cat([out_labels, out_labels_cr], 0)
    pos = all_labels.gt(thresh) if thresh < 1 else all_labels.eq(1)
    un = all_labels.lt(0)
    other = all_labels.lt(thresh).to(dtype)
    posf = pos.to(dtype)
    n_pos1 = (labels.gt(thresh) if thresh < 1 else labels.eq(1)).to(dtype).sum()
    n_neg = 2 * (n - n_pos1)
    sum_all = torch.empty(2 * n, device=device, dtype=dtype)
    sum_pos = torch.empty_like(sum_all)
    sum_oth = torch.empty_like(sum_all)
    e_pair = torch.empty_like(sum_all)
    idx = torch.arange(2 * n, device=device)
    for r0 in range(0, 2 * n, block):
        rows = idx[r0:r0 + block]
        s_ = tot[rows] @ tot.t() / T
        s_ = s_ - s_.max(dim=1, keepdim=True)[0]
        s_[torch.arange(rows.numel(), device=device), rows] = 0          # sims * (1 - eye) before the exp: the diagonal is exp(0)
        e = torch.exp(s_)
        sum_all[rows] = e.sum(1)
        sum_pos[rows] = (e * posf).sum(1)
        sum_oth[rows] = (e * other).sum(1)
        e_pair[rows] = e[torch.arange(rows.numel(), device=device), (rows + n) % (2 * n)]

    def calc_g(p, q, c):
        return torch.clamp((q - c * p) / (1 - c), min=np.e ** (-1 / T))

    pos_mean = sum_pos[pos] / (posf.sum() - 1)
    rem_mean = sum_oth[pos] / other.sum()
    sup = (-torch.log(pos_mean / (pos_mean + calc_g(pos_mean, rem_mean, tau_plus)))).mean()
    up = e_pair[un]
    urem = (sum_all[un] - up) / n_neg
    gp, gn = calc_g(up, urem, tau_plus), calc_g(up, urem, 1 - tau_plus)
    pr = preds[un]
    lpos = -torch.log(up / (up + gp)) * pr
    lneg = -torch.log(up / (up + gn)) * (1 - pr)
    unsup = torch.zeros((), device=device, dtype=dtype)
    hi, lo = pr.gt(0.99), pr.lt(0.01)
    mid = pr.gt(0.01) & pr.lt(0.99)
    if hi.any():
        unsup = unsup + lpos[hi].mean()
    if lo.any():
        unsup = unsup + lneg[lo].mean()
    if mid.any():
        unsup = unsup + lpos[mid].mean() + lneg[mid].mean()
    return sup.cpu(), unsup.cpu()


def _ucl_from_rowsums(sum_all, sum_pos, sum_oth, e_pair, preds, pos, un, other_sum, pos_sum, n_neg, T, tau_plus):
    """the scalar tail of `unbiased_con_loss*` (loss.py:644-694) as a function of the four row sums and the predictions"""
    def calc_g(p, q, c):
        return torch.clamp((q - c * p) / (1 - c), min=np.e ** (-1 / T))

    pos_mean = sum_pos[pos] / (pos_sum - 1)
    rem_mean = sum_oth[pos] / other_sum
    sup = (-torch.log(pos_mean / (pos_mean + calc_g(pos_mean, rem_mean, tau_plus)))).mean()
    up = e_pair[un]
    urem = (sum_all[un] - up) / n_neg
    gp, gn = calc_g(up, urem, tau_plus), calc_g(up, urem, 1 - tau_plus)
    pr = preds[un]
    lpos = -torch.log(up / (up + gp)) * pr
    lneg = -torch.log(up / (up + gn)) * (1 - pr)
    unsup = torch.zeros((), device=sum_all.device, dtype=sum_all.dtype)
    hi, lo = pr.gt(0.99), pr.lt(0.01)
    mid = pr.gt(0.01) & pr.lt(0.99)
    if hi.any():
        unsup = unsup + lpos[hi].mean()
    if lo.any():
        unsup = unsup + lneg[lo].mean()
    if mid.any():
        unsup = unsup + lpos[mid].mean() + lneg[mid].mean()
    return sup, unsup


class _StreamedUCL(torch.autograd.Function):
    """`unbiased_con_loss_streamed` WITH gradients, still without the (2N)^2 matrix: forward = the row sums, `block` rows at a
    time; backward = the chain rule through the scalar tail (autograd on 2N-vectors) and a second blocked pass
        d tot[i] = 1/T sum_{j != i} ( c(i; j) E_ij + c(j; i) E_ji ) tot[j],   E_ij = exp(S_ij - rowmax_i)  (rowmax detached, :55),
        c(i; j) = g_all[i] + g_pos[i] [pos j] + g_other[i] [other j] + g_pair[i] [j = pair(i)]
    (the masked diagonal is exp(0), a constant).  Pinned to autograd through the dense form by tests/test_oracle_losses.py."""

    @staticmethod
    def forward(ctx, f, f_cr, out_labels, out_labels_cr, labels, T, tau_plus, thresh, block, device, dtype):
        dev, dt = device or f.device, dtype or f.dtype
        n = f.shape[0]
        to = lambda t: t.detach().to(device=dev, dtype=dt)
        tot = torch.cat([to(f), to(f_cr)], 0)
        labels = to(labels)
        all_labels = torch.cat([labels, labels], 0)
        preds = torch.cat([to(out_labels), to(out_labels_cr)], 0)
        pos = all_labels.gt(thresh) if thresh < 1 else all_labels.eq(1)
        un = all_labels.lt(0)
        other = all_labels.lt(thresh).to(dt)
        posf = pos.to(dt)
        n_pos1 = (labels.gt(thresh) if thresh < 1 else labels.eq(1)).to(dt).sum()
        n_neg = 2 * (n - n_pos1)
        sums = [torch.empty(2 * n, device=dev, dtype=dt) for _ in range(4)]
        rowmax = torch.empty(2 * n, device=dev, dtype=dt)
        idx = torch.arange(2 * n, device=dev)
        for r0 in range(0, 2 * n, block):
            rows = idx[r0:r0 + block]
            ar = torch.arange(rows.numel(), device=dev)
            s_ = tot[rows] @ tot.t() / T
            rowmax[rows] = s_.max(dim=1)[0]
            s_ = s_ - rowmax[rows][:, None]
            s_[ar, rows] = 0
            e = torch.exp(s_)
            sums[0][rows] = e.sum(1)
            sums[1][rows] = (e * posf).sum(1)
            sums[2][rows] = (e * other).sum(1)
            sums[3][rows] = e[ar, (rows + n) % (2 * n)]
        with torch.enable_grad():
            leaves = [t.clone().requires_grad_() for t in sums + [preds]]
            sup, unsup = _ucl_from_rowsums(*leaves, pos, un, other.sum(), posf.sum(), n_neg, T, tau_plus)
        ctx.stuff = (tot, rowmax, leaves, sup, unsup, posf, other, n, T, block, f.device, f.dtype)
        return sup.detach().to(f.device, f.dtype), unsup.detach().to(f.device, f.dtype)

    @staticmethod
    def backward(ctx, g_sup, g_unsup):
        tot, rowmax, leaves, sup, unsup, posf, other, n, T, block, out_dev, out_dt = ctx.stuff
        dev, dt = tot.device, tot.dtype
        with torch.enable_grad():                           # (backward runs without grad mode; the scalar tail is a recorded graph)
            total = g_sup.detach().to(dev, dt) * sup + g_unsup.detach().to(dev, dt) * unsup
        ga, gp, go, gpair, gpreds = torch.autograd.grad(total, leaves, allow_unused=True)
        z = lambda g: torch.zeros(2 * n, device=dev, dtype=dt) if g is None else g
        ga, gp, go, gpair, gpreds = z(ga), z(gp), z(go), z(gpair), z(gpreds)
        dtot = torch.zeros_like(tot)
        idx = torch.arange(2 * n, device=dev)
        for r0 in range(0, 2 * n, block):
            rows = idx[r0:r0 + block]
            ar = torch.arange(rows.numel(), device=dev)
            pair = (rows + n) % (2 * n)
            e = torch.exp(tot[rows] @ tot.t() / T - rowmax[rows][:, None])
            w = e * (ga[rows][:, None] + gp[rows][:, None] * posf[None, :] + go[rows][:, None] * other[None, :])
            w[ar, pair] += gpair[rows] * e[ar, pair]
            w[ar, rows] = 0
            dtot[rows] += w @ tot / T
            dtot += w.t() @ tot[rows] / T
        back = lambda t: t.to(out_dev, out_dt)
        return (back(dtot[:n]), back(dtot[n:]), back(gpreds[:n]), back(gpreds[n:]), None, None, None, None, None, None, None)


def unbiased_con_loss_streamed_grad(labels, out_labels, out_labels_cr, f, f_cr, T, tau_plus, thresh, block=2048, device=None,
                                    dtype=None):
    """(sup, unsup) of `unbiased_con_loss_streamed`, differentiable w.r.t. f, f_cr, out_labels, out_labels_cr."""
    return _StreamedUCL.apply(f, f_cr, out_labels, out_labels_cr, labels, T, tau_plus, thresh, block, device, dtype)


def tomo_cr_semi_loss(hm_logits, hm_logits_cr, proj, proj_cr, gt, flip_prob, tau, temp, thresh, cr_weight, streamed=None):
    """`streamed`: None = the dense contrastive term (autograd works), or a dict of keyword arguments for
    `unbiased_con_loss_streamed` (values only; with "grad": True the differentiable blocked form `_StreamedUCL`)."""
    """trains/tomo_cr_semi_trainer.py:43-112, train phase with --contrastive.  Pinned since round 3 by
    tests/golden/semi_loss.npz (the reference's own TomoCRSemiLoss.forward, both flip branches, values and gradients;
    gen_golden.py::gen_semi_loss stubs the module's unused load-time imports)."""
    sig = lambda x: torch.clamp(torch.sigmoid(x), min=1e-4, max=1 - 1e-4)
    hm, hm_cr = sig(hm_logits), sig(hm_logits_cr)
    hm_loss = pu_neg_loss(hm, gt, tau)
    b, ch = proj.shape[:2]
    fd = -2 if flip_prob > 0.5 else -1
    pc, hc = proj_cr.flip(fd), hm_cr.flip(fd)
    f = proj.reshape(b, ch, -1).permute(1, 0, 2).reshape(ch, -1).T
    fc = pc.reshape(b, ch, -1).permute(1, 0, 2).reshape(ch, -1).T
    if streamed is not None and streamed.get("grad"):
        kw = {k: v for k, v in streamed.items() if k != "grad"}
        sup, unsup = unbiased_con_loss_streamed_grad(gt.reshape(-1), hm.reshape(-1), hc.reshape(-1), f, fc, temp, tau, thresh, **kw)
    elif streamed is not None:
        sup, unsup = unbiased_con_loss_streamed(gt.reshape(-1), hm.reshape(-1), hc.reshape(-1), f, fc, temp, tau, thresh, **streamed)
        sup, unsup = sup.to(hm.dtype), unsup.to(hm.dtype)
    else:
        sup, unsup = unbiased_con_loss(gt.reshape(-1), hm.reshape(-1), hc.reshape(-1), f, fc, temp, tau, thresh)
    cr = sup + 0.1 * unsup
    cons = mse(hm.reshape(-1), hc.reshape(-1))
    return hm_loss + cr * cr_weight + cons, hm_loss, cr, cons
