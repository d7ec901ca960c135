"""Mirror of the hot functions of cet_pick/utils/image.py (reference utils/image.py:42-193) on the
MI355X kernels: DoG particle picker, pooled NMS variants, greedy 3-D NMS.

Host arrays (numpy, as the reference passes them) are uploaded as fp32; GPU tensors are used in
place.  Results that the reference returns as numpy arrays are returned as numpy arrays.
"""
import ctypes

import numpy as np
import torch

from .. import _lib as L
from ..models import decode as _dec


def _to_dev(x, device=None):
    if isinstance(x, torch.Tensor):
        if not x.is_cuda:
            if not torch.cuda.is_available():
                raise L.HipExtensionError("no MI355X visible: the picker has no CPU path")
            x = x.to(device or "cuda")
        return x.to(torch.float32).contiguous()
    if not torch.cuda.is_available():
        raise L.HipExtensionError("no MI355X visible: the picker has no CPU path")
    return torch.as_tensor(np.ascontiguousarray(x, dtype=np.float32)).to(device or "cuda")


def gaussian_filter(vol, sigma):
    """scipy.ndimage.gaussian_filter(vol, sigma) (mode='reflect', truncate=4) for a (D,H,W) volume,
    fp32, as called at utils/image.py:152-156.  Returns a GPU tensor."""
    v = _to_dev(vol)
    d, h, w = v.shape
    out = torch.empty_like(v)
    tmp = torch.empty_like(v)
    L.check(L.lib().mi_gauss3d_sep(L.ptr(v), L.ptr(out), L.ptr(tmp), d, h, w, float(sigma), L.stream()),
            "mi_gauss3d_sep")
    return out


def non_maximum_suppression_3d(x, d, scale=1.0, threshold=-np.inf, max_out=None):
    """utils/image.py:42-79 (== models/decode.py:42-79).  Returns (scores, coords) numpy arrays,
    coords as (x, y, z) int32, in greedy (descending score) order."""
    v = _to_dev(x)
    D, H, W = v.shape
    lib = L.lib()
    ws = L.workspace(lib.mi_greedy_nms3d_workspace_bytes(D, H, W), v.device, "greedy")
    if max_out is None:
        max_out = min(D * H * W, 1 << 17)
    scores = torch.empty((max_out,), dtype=torch.float32, device=v.device)
    coords = torch.empty((max_out, 3), dtype=torch.int32, device=v.device)
    n = torch.zeros((1,), dtype=torch.int32, device=v.device)
    thr = float(threshold)
    if thr == -np.inf:
        thr = -3.4028234663852886e38  # every finite voxel passes, as in the reference
    L.check(lib.mi_greedy_nms3d(L.ptr(v), D, H, W, float(d), float(scale), thr, L.ptr(scores),
                                L.ptr(coords), L.ptr(n), max_out, L.ptr(ws), ws.numel(), L.stream()),
            "mi_greedy_nms3d")
    k = int(n.item())
    if k < 0:
        raise L.HipExtensionError("greedy NMS overflow (code %d): raise max_out / too many candidates" % k)
    return scores[:k].cpu().numpy(), coords[:k].cpu().numpy()


_nms_xy = _dec._nms_xy
_nms_z = _dec._nms_z


def _nms(heat, kernel=3):
    """utils/image.py:97-105: window (k,k,k) - NOT the (3,k,k) of models/decode.py."""
    return _dec._nms_generic(heat, kernel, kernel)


def _convert_1d_to_3d(inds, d, h, w):
    """utils/image.py:107-113: note `x = t % h` (the reference's own expression; equal to the
    decode.py variant whenever H == W)."""
    z_coord = torch.floor(inds.float() / (h * w)).int()
    t = inds.int() - (z_coord * h * w)
    y_coord = torch.floor(t.float() / w)
    x_coord = t % h
    return z_coord, y_coord, x_coord


def _topk(scores, K=900):
    """utils/image.py:115-125."""
    s, z, y, x, inds = _dec._topk(scores, K)
    batch, channel, depth, height, width = scores.size()
    z2, y2, x2 = _convert_1d_to_3d(inds, depth, height, width)
    return s, z2, y2, x2, inds


def dog_pick(rec, sigmas, kernel=3, border_z=10, nms_d=14, max_out=None, return_heat=False):
    """Device-resident form of `get_potential_coords_pyramid`: returns GPU tensors
    (scores (n,), coords (n,3) int32 x,y,z, cutoff ()) [+ dense NMS'd DoG heat-map]."""
    v = _to_dev(rec)
    D, H, W = v.shape
    lib = L.lib()
    ws = L.workspace(lib.mi_dog_pick_workspace_bytes(D, H, W, len(sigmas)), v.device, "dog")
    if max_out is None:
        max_out = min(D * H * W // 4 + 1024, 1 << 17)   # picks are >= d apart: 128 Ki covers a 512x512x512 volume at d=14
    scores = torch.empty((max_out,), dtype=torch.float32, device=v.device)
    coords = torch.empty((max_out, 3), dtype=torch.int32, device=v.device)
    n = torch.empty((1,), dtype=torch.int32, device=v.device)          # (both written by the chain on every path: the pick
    cutoff = torch.empty((1,), dtype=torch.float32, device=v.device)   # count / overflow code and the threshold)
    heat = torch.empty_like(v) if return_heat else None
    sig = (ctypes.c_float * len(sigmas))(*[float(s) for s in sigmas])
    L.check(lib.mi_dog_pick(L.ptr(v), D, H, W, ctypes.cast(sig, ctypes.c_void_p), len(sigmas),
                            int(kernel), int(border_z), int(nms_d), L.ptr(heat), L.ptr(scores),
                            L.ptr(coords), L.ptr(n), max_out, L.ptr(cutoff), L.ptr(ws), ws.numel(),
                            L.stream()), "mi_dog_pick")
    return scores, coords, n, cutoff, heat


def get_potential_coords_pyramid(rec, sigmas=[2, 4], num_pyramid=3, kernel=3, border_z=10):
    """utils/image.py:138-183.  (border_z is the reference's hard-coded 10 slices, exposed.)"""
    scores, coords, n, cutoff, _ = dog_pick(rec, sigmas, kernel=kernel, border_z=border_z)
    k = int(n.item())
    if k < 0:
        raise L.HipExtensionError("DoG picker overflow (code %d)" % k)
    return scores[:k].cpu().numpy(), coords[:k].cpu().numpy()


def get_potential_coords(rec, sigma1=2, sigma2=4, kernel=3, K=5000):
    """utils/image.py:185-193: DoG -> `_nms` (k,k,k) -> `_topk`."""
    v = _to_dev(rec)
    g1 = gaussian_filter(v, sigma1)
    g2 = gaussian_filter(v, sigma2)
    diff = (g2 - g1)[None, None]
    nms_diff = _nms(diff, kernel=kernel)
    topk_scores, topk_zs, topk_ys, topk_xs, topk_inds = _topk(nms_diff, K=K)
    return topk_zs, topk_ys, topk_xs
