"""Mirror of cet_pick/models/utils.py (hot-path part): `_sigmoid`, `_gather_feat`,
`_transpose_and_gather_feat` (reference models/utils.py:167-193)."""
import torch

from .. import _lib as L


class _SigmoidClampFn(torch.autograd.Function):
    """clamp(sigmoid(x), 1e-4, 1-1e-4) with its gradient s(1-s) inside the clamp, 0 outside (training losses)."""

    @staticmethod
    def forward(ctx, x):
        xs = x.detach().clone().contiguous()
        y = torch.empty_like(xs)
        L.check(L.lib().mi_sigmoid_clamp(L.ptr(xs), L.ptr(y), xs.numel(), L.stream()), "mi_sigmoid_clamp")
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        inside = (y > 1e-4) & (y < 1 - 1e-4)
        return torch.where(inside, dy * y * (1 - y), torch.zeros_like(dy))


def _sigmoid(x):
    """clamp(x.sigmoid_(), 1e-4, 1-1e-4): x is overwritten with the un-clamped sigmoid IN PLACE and
    a new clamped tensor is returned (reference models/utils.py:167-169).  When x carries a gradient the
    differentiable (out-of-place) form is used."""
    L.require_cuda(x, "x")
    if x.requires_grad and torch.is_grad_enabled():
        return _SigmoidClampFn.apply(x)
    if not x.is_contiguous():
        raise L.HipExtensionError("_sigmoid needs a contiguous tensor (it works in place)")
    y = torch.empty_like(x)
    L.check(L.lib().mi_sigmoid_clamp(L.ptr(x), L.ptr(y), x.numel(), L.stream()), "mi_sigmoid_clamp")
    return y


def _gather_feat(feat, ind, mask=None):
    # reference models/utils.py:171-182 - index plumbing on (N, K) indices, torch gather
    dim = feat.size(2)
    ind = ind.unsqueeze(2).expand(ind.size(0), ind.size(1), dim)
    feat = feat.gather(1, ind)
    if mask is not None:
        mask = mask.unsqueeze(2).expand_as(feat)
        feat = feat[mask]
        feat = feat.view(-1, dim)
    return feat


def _transpose_and_gather_feat(feat, ind):
    # reference models/utils.py:187-193
    feat = feat.permute(0, 2, 3, 4, 1).contiguous()
    feat = feat.view(feat.size(0), -1, feat.size(4))
    return _gather_feat(feat, ind)
