"""Tomogram loading on the MI355X: mirror of the reference's `cet_pick/utils/loader.py` (`quantize` :16-25,
`load_rec` :27-88, `preprocess` :90-121, `cutup` :124-132).

Same names and arguments; the arithmetic runs in `libcetpick_hip.so` (`csrc/preproc.hip`) and the results
stay on the device as fp32 tensors (the reference returns float64 numpy arrays; callers that need those use
`.cpu().numpy().astype(np.float64)`).  There is no CPU path.
"""
import math

import numpy as np
import torch

from .. import _lib as L
from . import mrc as _mrc

ORDERS = {"xyz": 0, "xzy": 1, "yxz": 2, "zxy": 3}
_MODE_OF = {np.dtype(np.int8): 0, np.dtype(np.int16): 1, np.dtype(np.float32): 2, np.dtype(np.uint16): 6}


def _dev():
    if not torch.cuda.is_available():
        raise L.HipExtensionError("cet_pick_amd.utils.loader needs the MI355X (cuda) device; there is no CPU path")
    return torch.device("cuda", torch.cuda.current_device())


def _stats(x, n_slices, slice_elems):
    lib = L.lib()
    stats = torch.empty((n_slices, 4), dtype=torch.float64, device=x.device)
    ws = L.workspace(lib.mi_vol_stats_workspace_bytes(n_slices, slice_elems), x.device, "volstats")
    L.check(lib.mi_vol_stats(L.ptr(x), n_slices, slice_elems, L.ptr(stats), L.ptr(ws), ws.numel(), L.stream()),
            "mi_vol_stats")
    return stats


def quantize(x, mi=-2.5, ma=2, dtype=np.uint8):
    """loader.py:16-25 on a device tensor: round(clip(255 (x - mi)/(ma - mi), 0, 255)) as uint8."""
    x = L.require_cuda(x, "x")
    mi = float(x.min()) if mi is None else mi
    ma = float(x.max()) if ma is None else ma
    q = torch.round(torch.clamp(255.0 * (x.double() - mi) / (ma - mi), 0, 255))
    return q.to(torch.uint8 if dtype == np.uint8 else torch.int32)


def rec_to_device(rec, order="xyz", compress=False):
    """The axis reorder (+ z-pair max) of load_rec for an in-memory MRC data block -> (Z', X, Y) fp32 cuda."""
    if order not in ORDERS:
        raise ValueError("order must be one of %s" % sorted(ORDERS))
    rec = np.asarray(rec)
    if rec.ndim != 3:
        raise ValueError("a 3-D MRC data block is required, got shape %s" % (rec.shape,))
    dt = np.dtype(rec.dtype).newbyteorder("=")
    if dt not in _MODE_OF:
        rec, dt = rec.astype(np.float32), np.dtype(np.float32)
    rec = np.ascontiguousarray(rec.astype(dt, copy=False))
    d0, d1, d2 = rec.shape
    zin = {"xyz": d2, "xzy": d1, "yxz": d2, "zxy": d0}[order]
    if order == "zxy" and compress and zin % 2:
        raise IndexError("zxy + compress needs an even number of slices (loader.py:64-75)")
    a, b = {"xyz": (d0, d1), "xzy": (d0, d2), "yxz": (d1, d0), "zxy": (d1, d2)}[order]
    zout = math.ceil(zin / 2) if compress else zin
    dev = _dev()
    src = torch.from_numpy(rec.view(np.uint8).reshape(-1)).to(dev, non_blocking=False)
    dst = torch.empty((zout, a, b), dtype=torch.float32, device=dev)
    L.check(L.lib().mi_rec_reorder(L.ptr(src), _MODE_OF[dt], d0, d1, d2, ORDERS[order], int(bool(compress)),
                                   L.ptr(dst), L.stream()), "mi_rec_reorder")
    return dst


def zscore(vol, per_slice=False):
    """(v - mean)/std over the whole volume (loader.py:59,:87) or slice by slice (is_tilt, :48-49)."""
    v = L.require_cuda(vol, "vol").contiguous()
    n, e = (v.shape[0], v[0].numel()) if per_slice else (1, v.numel())
    out = torch.empty_like(v)
    L.check(L.lib().mi_zscore(L.ptr(v), L.ptr(out), n, e, L.ptr(_stats(v, n, e)), L.stream()), "mi_zscore")
    return out


def load_rec(path, order="xyz", compress=False, is_tilt=False):
    """loader.py:27-88.  `path` may also be an already-read (nz, ny, nx) array."""
    rec = _mrc.open_data(path) if isinstance(path, (str, bytes)) or hasattr(path, "__fspath__") else path
    return zscore(rec_to_device(rec, order, compress), per_slice=is_tilt)


def preprocess(mrc, denoise=0, is_tilt=False):
    """loader.py:90-121: (Gaussian denoise) -> z-score -> 8-bit quantise -> min-max to [0, 1]."""
    v = mrc if isinstance(mrc, torch.Tensor) else torch.as_tensor(np.asarray(mrc), dtype=torch.float32)
    v = v.to(_dev(), torch.float32).contiguous()
    lib = L.lib()
    d, h, w = v.shape
    if denoise > 0:
        out, tmp = torch.empty_like(v), torch.empty_like(v)
        fn = lib.mi_gauss2d_slices if is_tilt else lib.mi_gauss3d_sep
        L.check(fn(L.ptr(v), L.ptr(out), L.ptr(tmp), d, h, w, float(denoise), L.stream()), "gauss")
        v = out
    n, e = (d, h * w) if is_tilt else (1, v.numel())
    # quantize() defaults (-2.5, 2) everywhere except the denoised volume branch (loader.py:105: -3, 3)
    mi, ma = (-3.0, 3.0) if (denoise > 0 and not is_tilt) else (-2.5, 2.0)
    out = torch.empty_like(v)
    L.check(lib.mi_zscore_quantize_minmax(L.ptr(v), L.ptr(out), n, e, L.ptr(_stats(v, n, e)), mi, ma,
                                          int(bool(is_tilt)), L.stream()), "mi_zscore_quantize_minmax")
    return out


def load_tomos_from_list(names, paths, order="xzy", compress=False, denoise=0, tilt=False):
    """loader.py:165-173: {name: preprocess(load_rec(path))} - device tensors in [0, 1]."""
    images = {}
    for name, path in zip(names, paths):
        im = load_rec(path, order=order, compress=compress, is_tilt=tilt)
        images[name] = preprocess(im, denoise=denoise, is_tilt=tilt)
    return images


def cutup(data, blck, strd):
    """loader.py:124-132: sliding windows (a strided view, no copy).  Device tensors and numpy arrays."""
    if isinstance(data, torch.Tensor):
        sh = np.array(data.shape)
        blck, strd = np.asanyarray(blck), np.asanyarray(strd)
        nbl = (sh - blck) // strd + 1
        st = np.array(data.stride())
        return data.as_strided(tuple(int(x) for x in np.r_[nbl, blck]), tuple(int(x) for x in np.r_[st * strd, st]))
    data = np.asarray(data)
    sh = np.array(data.shape)
    blck, strd = np.asanyarray(blck), np.asanyarray(strd)
    nbl = (sh - blck) // strd + 1
    return np.lib.stride_tricks.as_strided(data, shape=tuple(np.r_[nbl, blck]),
                                           strides=tuple(np.r_[np.array(data.strides) * strd, data.strides]))


def load_tlt(path):
    return np.loadtxt(path, ndmin=2)
