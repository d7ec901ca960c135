"""CPU: the loss oracle (oracle/loss_ref.py) against values and gradients produced by the reference's loss.py."""
import os

import numpy as np
import pytest
import torch

from oracle import loss_ref as O
from cet_pick_amd.synthetic import losses_inputs

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "losses.npz"))


def test_focal_pu_mse():
    pred, gt, f, f_cr, lab, o1, o2 = losses_inputs()
    p = pred.clone().requires_grad_()
    l = O.neg_loss(p, gt); l.backward()
    np.testing.assert_allclose(l.item(), G["focal"], rtol=1e-6)
    np.testing.assert_allclose(p.grad.numpy(), G["focal_grad"], rtol=1e-5, atol=1e-9)
    from cet_pick_amd.synthetic import confident_pred
    for tag, pr, tau in (("pu_0.05", pred, 0.05), ("pu_conf_0.6", confident_pred(gt), 0.6)):
        p = pr.clone().requires_grad_()
        l = O.pu_neg_loss(p, gt, tau); l.backward()
        np.testing.assert_allclose(l.item(), G[tag], rtol=1e-6)
        np.testing.assert_allclose(p.grad.numpy(), G[tag + "_grad"], rtol=1e-5, atol=1e-9)
    a = o1.clone().requires_grad_()
    l = O.mse(a, o2); l.backward()
    np.testing.assert_allclose(l.item(), G["mse"], rtol=1e-6)
    np.testing.assert_allclose(a.grad.numpy(), G["mse_grad"], rtol=1e-5, atol=1e-10)


def test_pu_branches_differ():
    # the random heat-map keeps the negative risk, the confident one drops it (neg_risk_total < 0): both pinned
    pred, gt = losses_inputs()[:2]
    assert G["pu_0.05_grad"][gt.numpy() == -1].any() and not G["pu_conf_0.6_grad"][gt.numpy() == -1].any()


@pytest.mark.parametrize("thresh", [1.0, 0.4])
def test_unbiased_con_loss(thresh):
    pred, gt, f, f_cr, lab, o1, o2 = losses_inputs()
    fa, fb = f.clone().requires_grad_(), f_cr.clone().requires_grad_()
    pa, pb = o1.clone().requires_grad_(), o2.clone().requires_grad_()
    sup, unsup = O.unbiased_con_loss(lab, pa, pb, fa, fb, 0.07, 0.03, thresh)
    (sup + 0.1 * unsup).backward()
    np.testing.assert_allclose(sup.item(), G[f"ucl_sup_{thresh}"], rtol=1e-5)
    np.testing.assert_allclose(unsup.item(), G[f"ucl_unsup_{thresh}"], rtol=1e-5)
    np.testing.assert_allclose(fa.grad.numpy(), G[f"ucl_gf_{thresh}"], rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(fb.grad.numpy(), G[f"ucl_gfcr_{thresh}"], rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(pa.grad.numpy(), G[f"ucl_gp_{thresh}"], rtol=1e-4, atol=1e-8)


@pytest.mark.parametrize("flip_prob", [0.2, 0.8])
def test_tomo_cr_semi_loss_matches_reference(flip_prob):
    """oracle == the reference's own TomoCRSemiLoss.forward (trains/tomo_cr_semi_trainer.py:43-112; semi_loss.npz)."""
    from cet_pick_amd.synthetic import semi_loss_inputs
    S = np.load(os.path.join(os.path.dirname(__file__), "golden", "semi_loss.npz"))
    gt, hm, hm_cr, pj, pj_cr = semi_loss_inputs(flip_prob)
    leaves = [t.clone().requires_grad_() for t in (hm, hm_cr, pj, pj_cr)]
    ref = O.tomo_cr_semi_loss(*leaves, gt, flip_prob, 0.1, 0.07, 0.5, 0.1)
    ref[0].backward()
    tag = "%.1f" % flip_prob
    for k, v in zip(("loss", "hm_loss", "cr_loss", "consis_loss"), ref):
        np.testing.assert_allclose(v.item(), S[f"{k}_{tag}"], rtol=2e-6)
    for name, t in zip(("g_hm", "g_hm_cr", "g_proj", "g_proj_cr"), leaves):
        g = t.grad.numpy()
        want = S[f"{name}_{tag}"]
        got = g if g.size < 4096 else g.reshape(-1)[::5]
        np.testing.assert_allclose(got, want, rtol=1e-4, atol=1e-6 * np.abs(want).max())
    # validation phase: the plain focal loss, no contrastive term
    sig = torch.clamp(torch.sigmoid(semi_loss_inputs(0.2)[1]), 1e-4, 1 - 1e-4)
    np.testing.assert_allclose(O.neg_loss(sig, semi_loss_inputs(0.2)[0]).item(), S["val_loss"], rtol=2e-6)
    assert S["val_cr_loss"] == 0


@pytest.mark.parametrize("thresh", [1.0, 0.4])
def test_streamed_contrastive_loss_equals_the_dense_form(thresh):
    """The row-blocked restatement (no (2N)^2 matrix; the oracle of the batch-16 step of C5) gives the dense oracle's
    values - itself pinned to the reference's UnbiasedConLoss two tests up - with ragged blocks, and in float64."""
    pred, gt, f, f_cr, lab, o1, o2 = losses_inputs()
    sup, unsup = O.unbiased_con_loss(lab, o1, o2, f, f_cr, 0.07, 0.03, thresh)
    for block in (37, 4096):
        s2, u2 = O.unbiased_con_loss_streamed(lab, o1, o2, f, f_cr, 0.07, 0.03, thresh, block=block)
        np.testing.assert_allclose(s2.item(), sup.item(), rtol=2e-6)
        np.testing.assert_allclose(u2.item(), unsup.item(), rtol=2e-6)
    d = lambda t: t.double()
    s64, u64 = O.unbiased_con_loss(d(lab), d(o1), d(o2), d(f), d(f_cr), 0.07, 0.03, thresh)
    s3, u3 = O.unbiased_con_loss_streamed(lab, o1, o2, f, f_cr, 0.07, 0.03, thresh, block=100, dtype=torch.float64)
    np.testing.assert_allclose(s3.item(), s64.item(), rtol=1e-10)
    np.testing.assert_allclose(u3.item(), u64.item(), rtol=1e-10)
    # through the step's loss: flags and flips as in tomo_cr_semi_loss
    from cet_pick_amd.synthetic import semi_loss_inputs
    gt4, hm, hm_cr, pj, pj_cr = semi_loss_inputs(0.8)
    a = O.tomo_cr_semi_loss(hm, hm_cr, pj, pj_cr, gt4, 0.8, 0.1, 0.07, 0.5, 0.1)
    b = O.tomo_cr_semi_loss(hm, hm_cr, pj, pj_cr, gt4, 0.8, 0.1, 0.07, 0.5, 0.1, streamed={"block": 500})
    for x, y in zip(a, b):
        np.testing.assert_allclose(y.item(), x.item(), rtol=2e-6)


@pytest.mark.parametrize("thresh", [1.0, 0.4])
def test_streamed_contrastive_loss_gradients_equal_autograd_through_the_dense_form(thresh):
    """`_StreamedUCL` (blocked forward, analytic blocked backward: the oracle of the C5 step's GRADIENTS at 16 pairs) against
    autograd through the dense oracle - which is pinned to the reference's gradients by test_unbiased_con_loss - in float64,
    with ragged blocks."""
    pred, gt, f, f_cr, lab, o1, o2 = losses_inputs()
    d = lambda t: t.double().clone().requires_grad_()
    A = [d(t) for t in (f, f_cr, o1, o2)]
    sup, unsup = O.unbiased_con_loss(lab.double(), A[2], A[3], A[0], A[1], 0.07, 0.03, thresh)
    (sup + 0.1 * unsup).backward()
    for block in (37, 4096):
        B = [d(t) for t in (f, f_cr, o1, o2)]
        s2, u2 = O.unbiased_con_loss_streamed_grad(lab.double(), B[2], B[3], B[0], B[1], 0.07, 0.03, thresh, block=block)
        (s2 + 0.1 * u2).backward()
        np.testing.assert_allclose(s2.item(), sup.item(), rtol=1e-10)
        np.testing.assert_allclose(u2.item(), unsup.item(), rtol=1e-10)
        for a, b in zip(A, B):
            np.testing.assert_allclose(b.grad.numpy(), a.grad.numpy(), rtol=1e-8, atol=1e-12 * float(a.grad.abs().max()) + 1e-300)
    # through the step's loss, flips included
    from cet_pick_amd.synthetic import semi_loss_inputs
    gt4, hm, hm_cr, pj, pj_cr = semi_loss_inputs(0.8)
    la = [t.double().clone().requires_grad_() for t in (hm, hm_cr, pj, pj_cr)]
    lb = [t.double().clone().requires_grad_() for t in (hm, hm_cr, pj, pj_cr)]
    O.tomo_cr_semi_loss(*la, gt4.double(), 0.8, 0.1, 0.07, 0.5, 0.1)[0].backward()
    O.tomo_cr_semi_loss(*lb, gt4.double(), 0.8, 0.1, 0.07, 0.5, 0.1, streamed={"block": 500, "grad": True})[0].backward()
    for a, b in zip(la, lb):
        np.testing.assert_allclose(b.grad.numpy(), a.grad.numpy(), rtol=1e-8, atol=1e-12)
