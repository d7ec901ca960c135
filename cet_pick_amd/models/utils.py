"""Mirror of cet_pick/models/utils.py (hot-path part): `_sigmoid`, `_gather_feat`,
`_transpose_and_gather_feat` (reference models/utils.py:167-193)."""
import torch

from .. import _lib as L


def _sigmoid(x):
    """clamp(x.sigmoid_(), 1e-4, 1-1e-4): x is overwritten with the un-clamped sigmoid IN PLACE and
    a new clamped tensor is returned (reference models/utils.py:167-169)."""
    L.require_cuda(x, "x")
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
