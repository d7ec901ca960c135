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
