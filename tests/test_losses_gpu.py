"""GPU parity of the detector-training losses (row a23) with the reference's values / gradients and the oracle."""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "losses.npz"))


def _inputs():
    from cet_pick_amd.synthetic import losses_inputs
    return [t.cuda() for t in losses_inputs()]


def test_focal_pu_mse_golden():
    from cet_pick_amd.models import loss as ML
    pred, gt, f, f_cr, lab, o1, o2 = _inputs()
    p = pred.clone().requires_grad_()
    l = ML.FocalLoss()(p, gt); (2.0 * l).backward()
    np.testing.assert_allclose(l.item(), G["focal"], rtol=2e-5)
    np.testing.assert_allclose(p.grad.cpu().numpy() / 2, G["focal_grad"], rtol=2e-4, atol=1e-7)
    from cet_pick_amd.synthetic import confident_pred
    for tag, pr, tau in (("pu_0.05", pred, 0.05), ("pu_conf_0.6", confident_pred(gt.cpu()).cuda(), 0.6)):
        p = pr.clone().requires_grad_()
        l = ML.PULoss(tau)(p, gt); l.backward()
        np.testing.assert_allclose(l.item(), G[tag], rtol=2e-5)
        np.testing.assert_allclose(p.grad.cpu().numpy(), G[tag + "_grad"], rtol=2e-4, atol=1e-7)
    a, b = o1.clone().requires_grad_(), o2.clone().requires_grad_()
    l = ML.ConsistencyLoss()(a, b); l.backward()
    np.testing.assert_allclose(l.item(), G["mse"], rtol=1e-5)
    np.testing.assert_allclose(a.grad.cpu().numpy(), G["mse_grad"], rtol=1e-5, atol=1e-9)
    np.testing.assert_allclose(b.grad.cpu().numpy(), -G["mse_grad"], rtol=1e-5, atol=1e-9)


def test_pu_without_positives_raises():
    from cet_pick_amd.models import loss as ML
    pred, gt = _inputs()[:2]
    with pytest.raises(ValueError):
        ML.PULoss(0.05)(pred, torch.where(gt == 1, torch.zeros_like(gt), gt))


@pytest.mark.parametrize("thresh", [1.0, 0.4])
def test_unbiased_con_loss_golden(thresh):
    from cet_pick_amd.models import loss as ML
    pred, gt, f, f_cr, lab, o1, o2 = _inputs()
    opt = SimpleNamespace(thresh=thresh, device=torch.device("cuda"))
    fa, fb = f.clone().requires_grad_(), f_cr.clone().requires_grad_()
    pa, pb = o1.clone().requires_grad_(), o2.clone().requires_grad_()
    sup, unsup = ML.UnbiasedConLoss(0.07, 0.03)(lab, pa, pb, fa, fb, opt)
    (sup + 0.1 * unsup).backward()
    np.testing.assert_allclose(sup.item(), G[f"ucl_sup_{thresh}"], rtol=1e-4)
    np.testing.assert_allclose(unsup.item(), G[f"ucl_unsup_{thresh}"], rtol=1e-4)
    for got, key in ((fa.grad, "ucl_gf"), (fb.grad, "ucl_gfcr"), (pa.grad, "ucl_gp"), (pb.grad, "ucl_gpcr")):
        ref = G[f"{key}_{thresh}"]
        np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=0, atol=2e-4 * np.abs(ref).max() + 1e-9)


@pytest.mark.parametrize("n,dim", [(1000, 32), (777, 16), (640, 64)])
def test_unbiased_con_loss_vs_oracle_ragged(n, dim):
    from oracle import loss_ref as O
    from cet_pick_amd.models import loss as ML
    from cet_pick_amd.synthetic import losses_inputs
    _, _, f, f_cr, lab, o1, o2 = losses_inputs(seed=n, n=n, dim=dim)
    ref_in = [t.clone().requires_grad_() for t in (f, f_cr, o1, o2)]
    sup_r, unsup_r = O.unbiased_con_loss(lab, ref_in[2], ref_in[3], ref_in[0], ref_in[1], 0.1, 0.05, 0.4)
    (sup_r + 0.1 * unsup_r).backward()
    dev_in = [t.clone().cuda().requires_grad_() for t in (f, f_cr, o1, o2)]
    opt = SimpleNamespace(thresh=0.4, device=torch.device("cuda"))
    sup, unsup = ML.UnbiasedConLoss(0.1, 0.05)(lab.cuda(), dev_in[2], dev_in[3], dev_in[0], dev_in[1], opt)
    (sup + 0.1 * unsup).backward()
    np.testing.assert_allclose(sup.item(), sup_r.item(), rtol=1e-4)
    np.testing.assert_allclose(unsup.item(), unsup_r.item(), rtol=1e-4)
    for a, b in zip(dev_in, ref_in):
        r = b.grad.numpy()
        np.testing.assert_allclose(a.grad.cpu().numpy(), r, rtol=0, atol=3e-4 * np.abs(r).max() + 1e-9)


def test_ucl_full_size_rowsums_properties():
    """N = 12,288 voxels per view (the 24,576^2 matrix is never built): row-sum identities."""
    from cet_pick_amd.models.loss import _UclRowSumsFn
    n2, dim = 24576, 32
    g = torch.Generator().manual_seed(0)
    f = torch.nn.functional.normalize(torch.randn(n2, dim, generator=g), dim=1).cuda()
    cls = torch.randint(0, 4, (n2,), generator=g).to(torch.uint8).cuda()
    m, sa, sp, so, ep = _UclRowSumsFn.apply(f, cls, 1.0 / 0.07)
    assert torch.allclose(m, torch.full_like(m, 1.0 / 0.07), rtol=1e-5)            # unit vectors: the diagonal is the maximum
    # sums against a blocked dense evaluation of 512 sampled rows
    rows = torch.arange(0, n2, 48, device="cuda")
    S = (f[rows] @ f.t()) / 0.07
    E = torch.exp(S - S.max(1, keepdim=True)[0])
    E[torch.arange(rows.numel()), rows] = 1.0
    np.testing.assert_allclose(sa[rows].cpu().numpy(), E.sum(1).cpu().numpy(), rtol=2e-4)
    np.testing.assert_allclose(sp[rows].cpu().numpy(), (E * (cls & 1).float()).sum(1).cpu().numpy(), rtol=2e-4, atol=1e-6)
    np.testing.assert_allclose(so[rows].cpu().numpy(), (E * ((cls >> 1) & 1).float()).sum(1).cpu().numpy(), rtol=2e-4, atol=1e-6)
    pair = (rows + n2 // 2) % n2
    np.testing.assert_allclose(ep[rows].cpu().numpy(), E[torch.arange(rows.numel()), pair].cpu().numpy(), rtol=2e-4, atol=1e-7)


@pytest.mark.parametrize("flip_prob", [0.2, 0.8])
def test_tomo_cr_semi_loss_vs_oracle(flip_prob):
    from oracle import loss_ref as O
    from cet_pick_amd.trains.tomo_cr_semi_trainer import TomoCRSemiLoss
    from cet_pick_amd.synthetic import semi_loss_inputs
    gt, hm, hm_cr, pj, pj_cr = semi_loss_inputs(flip_prob)
    ref_in = [t.clone().requires_grad_() for t in (hm, hm_cr, pj, pj_cr)]
    ref = O.tomo_cr_semi_loss(*ref_in, gt, flip_prob, 0.1, 0.07, 0.5, 0.1)
    ref[0].backward()
    opt = SimpleNamespace(pn=False, ge=False, tau=0.1, temp=0.07, thresh=0.5, cr_weight=0.1, num_stacks=1,
                          contrastive=True, device=torch.device("cuda"))
    dev_in = [t.clone().cuda().requires_grad_() for t in (hm, hm_cr, pj, pj_cr)]
    # `_sigmoid` works in place on a non-leaf (the network output)
    out = [{"hm": dev_in[0] * 1.0, "proj": dev_in[2]}]
    out_cr = [{"hm": dev_in[1] * 1.0, "proj": dev_in[3]}]
    loss, stats = TomoCRSemiLoss(opt)(out, {"hm": gt.cuda(), "flip_prob": flip_prob}, 1, "train", out_cr)
    loss.backward()
    for k, v in zip(("loss", "hm_loss", "cr_loss", "consis_loss"), ref):
        np.testing.assert_allclose(float(stats[k]), v.item(), rtol=2e-4)
    for a, b in zip(dev_in, ref_in):
        rr = b.grad.numpy()
        np.testing.assert_allclose(a.grad.cpu().numpy(), rr, rtol=0, atol=5e-4 * np.abs(rr).max() + 1e-9)
    # ... and against the reference's own TomoCRSemiLoss.forward (tests/golden/semi_loss.npz, gen_golden.py::gen_semi_loss)
    S = np.load(os.path.join(os.path.dirname(__file__), "golden", "semi_loss.npz"))
    tag = "%.1f" % flip_prob
    for k in ("loss", "hm_loss", "cr_loss", "consis_loss"):
        np.testing.assert_allclose(float(stats[k]), S[f"{k}_{tag}"], rtol=2e-4, err_msg=k)
    for name, a in zip(("g_hm", "g_hm_cr", "g_proj", "g_proj_cr"), dev_in):
        got = a.grad.cpu().numpy()
        got = got if got.size < 4096 else got.reshape(-1)[::5]
        want = S[f"{name}_{tag}"]
        np.testing.assert_allclose(got, want, rtol=0, atol=5e-4 * np.abs(want).max() + 1e-9, err_msg=name)
    val_loss, _ = TomoCRSemiLoss(opt)([{"hm": hm.cuda().clone(), "proj": None}], {"hm": gt.cuda()}, 1, "val")
    np.testing.assert_allclose(val_loss.item(), O.neg_loss(torch.clamp(torch.sigmoid(hm), 1e-4, 1 - 1e-4).squeeze(), gt.squeeze()).item(), rtol=1e-4)


@pytest.mark.parametrize("normalised", [True, False])
def test_ucl_backward_forms_agree(normalised, monkeypatch):
    """Round 6: the contrastive backward forms both terms of a similarity tile at once (S is symmetric: one product, one contraction) and,
    where the row maxima lie within 2^16 of each other, with ONE exponential per similarity - chosen on the device.  L2-normalised
    features take the one-exponential kernel, features whose norms spread the row maxima over 2^70 the general one; both must give the
    gradient of the two-launch form of rounds 2-5 (MI_UCL_BWD_SPLIT=1) and of a dense float64 evaluation."""
    from cet_pick_amd.models.loss import _UclRowSumsFn
    n2, dim, inv_T = 2048, 32, 1.0 / 0.07
    g = torch.Generator().manual_seed(11 + int(normalised))
    f = torch.nn.functional.normalize(torch.randn(n2, dim, generator=g), dim=1)
    if not normalised:
        f = f * torch.linspace(0.5, 2.0, n2)[torch.randperm(n2, generator=g)][:, None]
    cls = torch.randint(0, 4, (n2,), generator=g).to(torch.uint8).cuda()
    gouts = [torch.randn(n2, generator=g).cuda() * s for s in (0.0, 1.0, 0.7, 0.5, 0.3)]       # (no gradient through the maximum itself)

    def grad():
        fa = f.clone().cuda().requires_grad_()
        outs = _UclRowSumsFn.apply(fa, cls, inv_T)
        torch.autograd.backward(list(outs[1:]), [go.clone() for go in gouts[1:]])
        return fa.grad.clone(), [o.detach() for o in outs]

    g_default, outs = grad()
    spread = float((outs[0].max() - outs[0].min()) * 1.4426950408889634)
    assert (spread <= 16.0) == normalised, spread
    monkeypatch.setenv("MI_UCL_BWD_TWO_EXP", "1")
    g_two, _ = grad()
    monkeypatch.delenv("MI_UCL_BWD_TWO_EXP")
    monkeypatch.setenv("MI_UCL_BWD_SPLIT", "1")
    g_split, _ = grad()
    scale = float(g_split.abs().max())
    if not normalised:
        assert torch.equal(g_default, g_two)               # the device picked the general kernel: same launch
    assert float((g_default - g_split).abs().max()) <= 2e-5 * scale
    assert float((g_two - g_split).abs().max()) <= 2e-5 * scale
    # dense float64: E = exp(S - rowmax) off the diagonal, row sums weighted by the class masks, the pair element
    F64 = f.double().cuda().requires_grad_()
    S = (F64 @ F64.t()) * inv_T
    m = S.max(1, keepdim=True)[0].detach()
    E = torch.exp(S - m) * (1 - torch.eye(n2, dtype=torch.float64, device="cuda"))
    pos, oth = (cls & 1).double(), ((cls >> 1) & 1).double()
    pair = (torch.arange(n2, device="cuda") + n2 // 2) % n2
    sa, sp, so = E.sum(1) + 1, (E * pos).sum(1) + pos, (E * oth).sum(1) + oth
    ep = torch.exp(S - m)[torch.arange(n2), pair]
    (sa * gouts[1].double() + sp * gouts[2].double() + so * gouts[3].double() + ep * gouts[4].double()).sum().backward()
    ref = F64.grad
    assert float((g_default.double() - ref).abs().max()) <= 1e-4 * float(ref.abs().max())
