"""Sub-tomogram extraction on the GPU: the crop + normalise arithmetic of
the reference datasets, without their per-pick Python loops.

`extract_subvols` / `extract_subvols_3d` follow datasets/tomo_pre_proj_angle_select_new3d_vol.py:117-138
for a whole list of picks at once; `crop_znorm` makes the z-normalised 3-D crops the MoCo-3D encoder
consumes.  The volume stays resident on the device; centres are (x, y, z) like the picker returns.
"""
import numpy as np
import torch

from .. import _lib as L

RAW, SUMZ_MINMAX, ZNORM = 0, 1, 2


def _centres(c, device):
    if isinstance(c, torch.Tensor):
        return c.to(device=device, dtype=torch.int32).contiguous()
    return torch.as_tensor(np.ascontiguousarray(c, dtype=np.int32)).to(device)


def _crop(vol, centres, size, mode, flip_x=False):
    L.require_cuda(vol, "vol")
    vol = vol.contiguous()
    d, h, w = vol.shape
    cz, cy, cx = [int(v) for v in size]
    c = _centres(centres, vol.device)
    n = c.shape[0]
    shape = (n, cy, cx) if mode == SUMZ_MINMAX else (n, cz, cy, cx)
    out = torch.empty(shape, dtype=torch.float32, device=vol.device)
    L.check(L.lib().mi_crop_normalize(L.ptr(vol), d, h, w, L.ptr(c), n, cz, cy, cx, mode, int(bool(flip_x)),
                                      L.ptr(out), L.stream()), "mi_crop_normalize")
    return out


def extract_subvols(v, tomo_coords, subvol_size):
    """:117-128 for every pick: (n, 1, sy, sx) float32 = min-max(sum_z v[z-sz//2 : z+sz//2+1, ...]).
    subvol_size = (sz, sy, sx); an odd sz gives the z-1..z+1 slab of the reference."""
    sz, sy, sx = subvol_size
    return _crop(v, tomo_coords, (2 * (int(sz) // 2) + 1, sy, sx), SUMZ_MINMAX).unsqueeze(1)


def extract_subvols_3d(v, tomo_coords, subvol_size):
    """:130-138 for every pick: raw (n, 2*(sz//2)+1, sy, sx) crops."""
    sz, sy, sx = subvol_size
    return _crop(v, tomo_coords, (2 * (int(sz) // 2) + 1, sy, sx), RAW)


def crop_znorm(v, tomo_coords, size, flip_x=False):
    """(n, 1, cz, cy, cx) z-normalised crops (mean 0, unbiased std 1), optionally mirrored along x."""
    return _crop(v, tomo_coords, size, ZNORM, flip_x).unsqueeze(1)
