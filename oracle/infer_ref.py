"""Oracle for the inference path: numpy float64/float32 on the CPU.

TEST INFRASTRUCTURE ONLY - see oracle/__init__.py.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


# --------------------------------------------------------------------------- a20
def sigmoid_clamp(x):
    """models/utils.py:167-169 `_sigmoid`: clamp(sigmoid(x), 1e-4, 1-1e-4), fp32."""
    x = np.asarray(x, dtype=np.float32)
    y = (1.0 / (1.0 + np.exp(-x.astype(np.float64)))).astype(np.float32)
    return np.clip(y, np.float32(1e-4), np.float32(1 - 1e-4))


# --------------------------------------------------------------------------- a16
def _pool_max(v, win):
    """Stride-1 max-pool of a (D,H,W) array with -inf padding (torch max_pool3d semantics)."""
    out = v
    for axis, k in enumerate(win):
        if k == 1:
            continue
        p = (k - 1) // 2
        pad = [(0, 0)] * 3
        pad[axis] = (p, p)
        vp = np.pad(out, pad, mode="constant", constant_values=-np.inf)
        acc = None
        n = out.shape[axis]
        for s in range(k):
            sl = [slice(None)] * 3
            sl[axis] = slice(s, s + n)
            cur = vp[tuple(sl)]
            acc = cur if acc is None else np.maximum(acc, cur)
        out = acc
    return out


def nms_window(heat, win):
    """keep = (max_pool3d(heat, win, stride 1, pad (k-1)//2) == heat); return heat*keep.

    decode.py:11-33: `_nms_xy` win=(1,k,k), `_nms_z` win=(k,1,1), `_nms` win=(3,k,k);
    utils/image.py:97-105 `_nms` win=(k,k,k).  heat: (D,H,W)."""
    heat = np.asarray(heat)
    hmax = _pool_max(heat, win)
    return heat * (hmax == heat).astype(heat.dtype)


# --------------------------------------------------------------------------- a17/a18
def convert_1d_to_3d(inds, d, h, w, image_variant=False):
    """decode.py:35-41 (float32 division!) / image.py:107-113 (x = t % h)."""
    inds = np.asarray(inds, dtype=np.int64)
    z = np.floor(inds.astype(np.float32) / np.float32(h * w)).astype(np.int32)
    t = inds.astype(np.int32) - z * np.int32(h * w)
    y = np.floor(t.astype(np.float32) / np.float32(w))
    x = t % (h if image_variant else w)
    return z, y.astype(np.float32), x.astype(np.int32)


def topk(scores, K):
    """decode.py:82-92 `_topk` for batch=channel=1.  Ties: lowest flat index first (documented
    build choice; torch.topk leaves tie order unspecified)."""
    flat = np.asarray(scores).reshape(-1)
    order = np.lexsort((np.arange(flat.size), -flat.astype(np.float64)))[:K]
    d, h, w = scores.shape
    z, y, x = convert_1d_to_3d(order, d, h, w)
    return flat[order], z, y, x, order


def tomo_decode(heat, kernel=3, K=900, if_fiber=False):
    """decode.py:123-155 with reg=None, batch=cat=1.  heat (D,H,W) -> (K,5) [x+.25,y+.25,z,s,s]."""
    if if_fiber:
        h = nms_window(heat, (1, kernel, kernel))
        h = nms_window(h, (kernel, 1, 1))
    else:
        h = nms_window(heat, (3, kernel, kernel))
    s, z, y, x, _ = topk(h, K)
    out = np.stack([x.astype(np.float32) + np.float32(0.25), y + np.float32(0.25),
                    z.astype(np.float32), s.astype(np.float32), s.astype(np.float32)], axis=1)
    return out.astype(np.float32)


# --------------------------------------------------------------------------- gaussian (scipy restated)
def gaussian_kernel1d(sigma, truncate=4.0):
    """scipy.ndimage `_gaussian_kernel1d` (order 0): radius=int(truncate*sigma+0.5), normalised."""
    radius = int(truncate * float(sigma) + 0.5)
    x = np.arange(-radius, radius + 1, dtype=np.float64)
    phi = np.exp(-0.5 / (float(sigma) * float(sigma)) * x ** 2)
    return phi / phi.sum(), radius


def reflect_index(i, n):
    """scipy mode='reflect' (d c b a | a b c d | d c b a): half-sample symmetric."""
    i = np.asarray(i)
    period = 2 * n
    i = np.mod(i, period)
    return np.where(i >= n, period - 1 - i, i)


def gaussian_filter(vol, sigma, dtype=np.float64):
    """Separable Gaussian, axis order 0,1,2 as scipy.ndimage.gaussian_filter (utils/image.py:152-156
    calls it on the float64 (Z,H,W) volume)."""
    out = np.asarray(vol, dtype=dtype)
    w, r = gaussian_kernel1d(sigma)
    w = w.astype(dtype)
    for axis in range(3):
        n = out.shape[axis]
        idx = reflect_index(np.arange(-r, n + r), n)
        src = np.take(out, idx, axis=axis)
        acc = np.zeros_like(out)
        for t in range(2 * r + 1):
            sl = [slice(None)] * 3
            sl[axis] = slice(t, t + n)
            acc += w[t] * src[tuple(sl)]
        out = acc
    return out


# --------------------------------------------------------------------------- a19
_greedy_lib = None


def _load_greedy():
    global _greedy_lib
    if _greedy_lib is None:
        so = os.path.join(_HERE, "libgreedy_ref.so")
        src = os.path.join(_HERE, "greedy_nms.c")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-o", so, src])
        _greedy_lib = ctypes.CDLL(so)
        _greedy_lib.greedy_nms3d_ref.restype = ctypes.c_long
    return _greedy_lib


def ball_deltas(shape, d, scale=1.0):
    """decode.py:43-54: flat-index offsets of the radius r=scale*d/2 ball (no bounds check)."""
    r = scale * d / 2
    width = int(np.ceil(r))
    A = np.arange(-width, width + 1)
    ii, jj, kk = np.meshgrid(A, A, A)
    mask = (ii ** 2 + jj ** 2 + kk ** 2) <= r * r
    return (ii[mask] * (shape[1] * shape[2]) + jj[mask] * shape[2] + kk[mask]).astype(np.int64)


def non_maximum_suppression_3d(x, d, scale=1.0, threshold=-np.inf):
    """decode.py:42-79 == image.py:42-79.  Visit voxels in np.argsort(...)[::-1] order; an
    unsuppressed voxel above threshold becomes a pick and suppresses i+delta for every delta of the
    ball, as FLAT offsets (wraps across rows/slices; out-of-range offsets are inert)."""
    x = np.ascontiguousarray(x)
    A = x.ravel()
    order = np.ascontiguousarray(np.argsort(A, axis=None)[::-1].astype(np.int64))
    deltas = np.ascontiguousarray(ball_deltas(x.shape, d, scale))
    A64 = np.ascontiguousarray(A.astype(np.float64))
    n = A.size
    picks = np.zeros(n, dtype=np.int64)
    supp = np.zeros(n, dtype=np.uint8)
    lib = _load_greedy()
    thr = float(threshold)
    j = lib.greedy_nms3d_ref(
        A64.ctypes.data_as(ctypes.c_void_p), order.ctypes.data_as(ctypes.c_void_p),
        ctypes.c_long(n), deltas.ctypes.data_as(ctypes.c_void_p), ctypes.c_long(deltas.size),
        ctypes.c_double(thr), supp.ctypes.data_as(ctypes.c_void_p),
        picks.ctypes.data_as(ctypes.c_void_p))
    picks = picks[:j]
    zz, yy, xx = np.unravel_index(picks, x.shape)
    scores = A[picks].astype(np.float32)
    coords = np.stack([xx, yy, zz], axis=1).astype(np.int32).reshape(-1, 3)
    return scores, coords


# --------------------------------------------------------------------------- a14
def dog_nms_heat(rec, sigmas, kernel=3, border_z=10, dtype=np.float64):
    """utils/image.py:138-176: DoG levels -> border zero -> `_nms_xy` -> max over levels."""
    rec = np.asarray(rec, dtype=dtype)
    z, r, c = rec.shape
    bx = by = 30
    if r > 512 and c > 512:
        bx, by = 60, 60
    ims = [gaussian_filter(rec, s, dtype=dtype) for s in sigmas]
    out = None
    for i in range(len(sigmas) - 1):
        diff = ims[i + 1] - ims[i]
        if border_z > 0:
            diff[:border_z] = 0
            diff[-border_z:] = 0
        diff[:, :bx, :] = 0
        diff[:, -bx:, :] = 0
        diff[:, :, :by] = 0
        diff[:, :, -by:] = 0
        n = nms_window(diff, (1, kernel, kernel))
        out = n if out is None else np.maximum(out, n)
    return out


def pos_threshold(heat):
    """image.py:177-179: mean(pos) + 0.5*std(pos) (torch .std() is the unbiased estimator)."""
    pos = heat[heat > 0].astype(np.float64)
    return float(pos.mean() + 0.5 * pos.std(ddof=1))


def get_potential_coords_pyramid(rec, sigmas=(2, 4), kernel=3, border_z=10, nms_d=14,
                                 dtype=np.float64):
    """utils/image.py:138-183 -> (scores f32 (n,), coords i32 (n,3) as x,y,z)."""
    heat = dog_nms_heat(rec, sigmas, kernel, border_z, dtype)
    cutoff = pos_threshold(heat)
    return non_maximum_suppression_3d(heat, nms_d, threshold=cutoff)


# --------------------------------------------------------------------------- a13
# Pinned by tests/golden/crops.npz (the reference's own methods, imported by tests/golden/gen_golden.py:gen_crops), except
# the torchio chain (znorm_rescale_znorm) and the torchvision 8-bit round trip (u8_roundtrip_normalize): those two
# packages are absent here (un-vendored) - their published arithmetic is restated: parity unpinned.
def extract_subvols(v, coord, size):
    """datasets/tomo_pre_proj_angle_select_new3d_vol.py:117-128 (one pick, numpy like the reference)."""
    sz, sy, sx = size
    x, y, z = coord
    sub = v[z - sz // 2:z + sz // 2 + 1, y - sy // 2:y + sy // 2, x - sx // 2:x + sx // 2].copy()
    sub = np.sum(sub, axis=0)
    sub = (sub - np.min(sub)) / (np.max(sub) - np.min(sub))
    return sub.astype(np.float32)[None]


def extract_subvols_3d(v, coord, size):
    """...:130-138."""
    sz, sy, sx = size
    x, y, z = coord
    return v[z - sz // 2:z + sz // 2 + 1, y - sy // 2:y + sy // 2, x - sx // 2:x + sx // 2].copy()


def extract_3d_tomo(rec, coord, crop_x, crop_y):
    """...:109-115: one slice, min-max."""
    x, y, z = coord
    p = rec[z, y - crop_y // 2:y + crop_y // 2, x - crop_x // 2:x + crop_x // 2].copy()
    p = (p - np.min(p)) / (np.max(p) - np.min(p))
    return p.astype(np.float32)[None]


def subvol_mean_std(subvols):
    """...:238-239: torch.mean / torch.std (unbiased) over the stack of all crops."""
    a = np.asarray(subvols, dtype=np.float64)
    return float(a.mean()), float(a.std(ddof=1))


def cutup(data, blck, strd):
    """utils/loader.py:124-132: sliding windows as a strided view."""
    data = np.asarray(data)
    sh = np.array(data.shape)
    blck, strd = np.asanyarray(blck), np.asanyarray(strd)
    nbl = (sh - blck) // strd + 1
    return np.lib.stride_tricks.as_strided(data, shape=tuple(np.r_[nbl, blck]),
                                           strides=tuple(np.r_[np.array(data.strides) * strd, data.strides]))


def znorm_rescale_znorm(window, margin):
    """datasets/tomo_pre.py:57-60 on one cutup window (the deterministic tail of the torchio chain; torchio 0.18 semantics
    restated - parity unpinned): Crop(margin per side) -> ZNormalization (mean, unbiased std of all voxels) ->
    RescaleIntensity(out_min_max=(-3, 3), percentiles (0, 100)) -> ZNormalization."""
    mz, my, mx = margin
    w = np.asarray(window, dtype=np.float32)
    c = w[mz:w.shape[0] - mz, my:w.shape[1] - my, mx:w.shape[2] - mx].astype(np.float64)
    c = (c - c.mean()) / c.std(ddof=1)
    lo, hi = c.min(), c.max()
    c = (c - lo) / (hi - lo) * 6.0 - 3.0
    c = (c - c.mean()) / c.std(ddof=1)
    return c.astype(np.float32)


def u8_roundtrip_normalize(sub, mean, std):
    """simsiam_test_hm_3d.py:45-51: T.ToPILImage() (float tensor: mul(255).byte(), i.e. truncation) -> T.ToTensor()
    (/255) -> T.Normalize(mean, std).  torchvision 0.12 semantics restated - parity unpinned."""
    q = np.floor(np.asarray(sub, dtype=np.float32) * np.float32(255.0)).astype(np.uint8).astype(np.float32) / np.float32(255.0)
    return ((q - np.float32(mean)) / np.float32(std)).astype(np.float32)
