"""Mirror of cet_pick/models/decode.py on the MI355X kernels.

Same names and argument meaning as the reference (models/decode.py:11-155).  Tensors must live on
the GPU; every function launches hand-written HIP through the C-ABI (include/cetpick_hip.h).
"""
import numpy as np
import torch

from .. import _lib as L
from .utils import _transpose_and_gather_feat


def _nms_generic(heat, kd, kh):
    L.require_cuda(heat, "heat")
    if heat.dim() != 5:
        raise ValueError("heat must be (B, C, D, H, W)")
    heat = heat.contiguous()
    out = torch.empty_like(heat)
    b, c, d, h, w = heat.shape
    fn = L.lib().mi_nms3d
    for i in range(b * c):
        src = heat.view(b * c, d, h, w)[i]
        dst = out.view(b * c, d, h, w)[i]
        L.check(fn(L.ptr(src), L.ptr(dst), d, h, w, kd, kh, L.stream()), "mi_nms3d")
    return out


def _nms_xy(heat, kernel=3):
    """decode.py:11-17: max-pool window (1,k,k)."""
    return _nms_generic(heat, 1, kernel)


def _nms_z(heat, kernel=3):
    """decode.py:19-25: max-pool window (k,1,1)."""
    return _nms_generic(heat, kernel, 1)


def _nms(heat, kernel=3):
    """decode.py:27-33: max-pool window (3,k,k)."""
    return _nms_generic(heat, 3, kernel)


def _convert_1d_to_3d(inds, d, h, w):
    """decode.py:35-41 (float32 division kept on purpose)."""
    z_coord = torch.floor(inds.float() / (h * w)).int()
    t = inds.int() - (z_coord * h * w)
    y_coord = torch.floor(t.float() / w)
    x_coord = t % w
    return z_coord, y_coord, x_coord


def _decode_fused(vol, kernel, K, fiber, apply_sigmoid, heat_out, dets=None):
    """One volume through mi_sigmoid_nms_topk; `dets` (K,5) is written in place when given (a row of the batch's
    output: no stack / copy kernel afterwards)."""
    d, h, w = vol.shape
    lib = L.lib()
    ws = _decode_workspace(lib.mi_decode_workspace_bytes(d, h, w, K), vol.device)
    if dets is None:
        dets = torch.empty((K, 5), dtype=torch.float32, device=vol.device)
    nvalid = torch.empty((1,), dtype=torch.int32, device=vol.device)
    rc = lib.mi_sigmoid_nms_topk(L.ptr(vol), L.ptr(heat_out), d, h, w, kernel, int(bool(fiber)),
                                 int(bool(apply_sigmoid)) | _WS_CLEAN, K, L.ptr(dets), L.ptr(nvalid),
                                 L.ptr(ws), ws.numel(), L.stream())
    if rc:
        L.drop_workspace(vol.device, "decode")
    L.check(rc, "mi_sigmoid_nms_topk")
    return dets, nvalid


_WS_CLEAN = 2       # include/cetpick_hip.h: bit 1 of `apply_sigmoid` - the header of this workspace is kept clean


def _decode_workspace(nbytes, device):
    """The decode workspace, its header zeroed once per allocation (mi_decode_workspace_init); every call made with
    _WS_CLEAN leaves it zeroed again, so a decode is march + filter + final and no clearing pass."""
    def init(buf):
        L.check(L.lib().mi_decode_workspace_init(L.ptr(buf), buf.numel(), L.stream()), "mi_decode_workspace_init")
    return L.workspace(nbytes, device, "decode", init=init)


def tomo_decode(heat, kernel=3, reg=None, K=900, if_fiber=False):
    """decode.py:123-155.  heat: (B, 1, D, H, W) already sigmoid'ed -> (B, K, 5) = [x, y, z, s, s]."""
    L.require_cuda(heat, "heat")
    batch, cat, depth, height, width = heat.size()
    if cat != 1:
        raise ValueError("tomo_decode expects one heat-map channel")
    heat = heat.contiguous()
    detections = torch.empty((batch, K, 5), dtype=torch.float32, device=heat.device)
    for b in range(batch):
        _decode_fused(heat[b, 0], kernel, K, if_fiber, False, None, detections[b])
    if reg is not None:
        # decode.py:134-140: sub-voxel offsets gathered at the peak indices
        xs = (detections[:, :, 0] - 0.25).long()
        ys = (detections[:, :, 1] - 0.25).long()
        zs = detections[:, :, 2].long()
        inds = (zs * height + ys) * width + xs
        r = _transpose_and_gather_feat(reg, inds).view(batch, K, 2)
        detections = detections.clone()
        detections[:, :, 0] = xs.float() + r[:, :, 0]
        detections[:, :, 1] = ys.float() + r[:, :, 1]
    return detections


def sigmoid_tomo_decode(logits, kernel=3, K=900, if_fiber=False):
    """Fused `_sigmoid` + `tomo_decode` (detectors/tomo_det.py:33-37 calls them back to back):
    one pass over the logits produces the clamped heat-map and the (B, K, 5) detections."""
    L.require_cuda(logits, "logits")
    batch, cat, depth, height, width = logits.size()
    logits = logits.contiguous()
    heat = torch.empty_like(logits)
    detections = torch.empty((batch, K, 5), dtype=torch.float32, device=logits.device)
    for b in range(batch):
        _decode_fused(logits[b, 0], kernel, K, if_fiber, True, heat[b, 0], detections[b])
    return heat, detections


def _topk(scores, K=900):
    """decode.py:82-92 for the (B,1,D,H,W) case through the same selection kernels."""
    L.require_cuda(scores, "scores")
    batch, channel, depth, height, width = scores.size()
    if channel != 1:
        raise ValueError("_topk expects one channel")
    scores = scores.contiguous()
    s_l, z_l, y_l, x_l, i_l = [], [], [], [], []
    for b in range(batch):
        d, h, w = depth, height, width
        lib = L.lib()
        ws = _decode_workspace(lib.mi_decode_workspace_bytes(d, h, w, K), scores.device)
        dets = torch.empty((K, 5), dtype=torch.float32, device=scores.device)
        # window (1,1,1): every positive voxel is its own maximum -> plain top-K
        vol = scores[b, 0]
        rc = _topk_plain(vol, K, dets, ws)
        L.check(rc, "mi_sigmoid_nms_topk")
        xs = (dets[:, 0] - 0.25).int()
        ys = dets[:, 1] - 0.25
        zs = dets[:, 2].int()
        s_l.append(dets[:, 3]); z_l.append(zs); y_l.append(ys); x_l.append(xs)
        i_l.append((zs.long() * h + ys.long()) * w + xs.long())
    return (torch.stack(s_l).view(batch, 1, K), torch.stack(z_l), torch.stack(y_l),
            torch.stack(x_l), torch.stack(i_l))


def _topk_plain(vol, K, dets, ws):
    d, h, w = vol.shape
    # fiber mode with k=1 pools nothing (windows (1,1,1) then (1,1,1))
    rc = L.lib().mi_sigmoid_nms_topk(L.ptr(vol), None, d, h, w, 1, 1, _WS_CLEAN, K, L.ptr(dets), None,
                                     L.ptr(ws), ws.numel(), L.stream())
    if rc:
        L.drop_workspace(vol.device, "decode")
    return rc


def non_maximum_suppression_3d(x, d, scale=1.0, threshold=-np.inf):
    """decode.py:42-79.  x: (D,H,W) GPU tensor -> (scores f32 (n,), coords i32 (n,3) as x,y,z),
    returned as numpy arrays like the reference."""
    from ..utils.image import non_maximum_suppression_3d as impl
    return impl(x, d, scale=scale, threshold=threshold)
