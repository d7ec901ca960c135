"""Sub-tomogram extraction on the GPU: the crop + normalise arithmetic of
the reference datasets, without their per-pick Python loops.

`extract_subvols` / `extract_subvols_3d` follow datasets/tomo_pre_proj_angle_select_new3d_vol.py:117-138
for a whole list of picks at once; `crop_znorm` makes the z-normalised 3-D crops the MoCo-3D encoder
consumes.  The volume stays resident on the device; centres are (x, y, z) like the picker returns.
"""
import numpy as np
import torch

from .. import _lib as L

RAW, SUMZ_MINMAX, ZNORM, ZNORM_RESCALE_ZNORM = 0, 1, 2, 3


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


class CropTable:
    """Everything `mi_crop_normalize_table` needs to cut any batch of a dataset, resident on the device: one descriptor
    per tomogram (pointer + extents), and per sample its tomogram, its centre (x, y, z) and the second view's shift.
    Built once per dataset; a batch is then `cut(order, first, n, ...)` - one launch per view, no host arithmetic, no
    host-to-device copy (the per-batch work of the reference's DataLoader workers, datasets/particle_pre_3d_vol.py:70-85)."""

    def __init__(self, vols, owner, centres, shift=None):
        dev = vols[0].device
        self.vols = [L.require_cuda(v, "vol").contiguous() for v in vols]          # (kept alive: the table holds raw pointers)
        for v in self.vols:
            if v.dtype != torch.float32 or v.dim() != 3:
                raise L.HipExtensionError("CropTable: tomograms must be (D, H, W) float32 device tensors")
        desc = np.zeros((len(self.vols), 3), dtype=np.int64)                          # struct mi_vol_desc {ptr; D, H; W, 0}
        for i, v in enumerate(self.vols):
            d, h, w = (int(a) for a in v.shape)
            desc[i] = (v.data_ptr(), d | (h << 32), w)
        self.desc = torch.as_tensor(desc).to(dev)
        self.owner = torch.as_tensor(np.ascontiguousarray(owner, dtype=np.int32)).to(dev)
        self.centres = _centres(centres, dev)
        self.shift = _centres(shift, dev) if shift is not None else None
        self.n = int(self.centres.shape[0])
        if int(self.owner.numel()) != self.n or (self.shift is not None and tuple(self.shift.shape) != tuple(self.centres.shape)):
            raise L.HipExtensionError("CropTable: owner / centres / shift disagree in length")

    def epoch_order(self, order):
        """the epoch's sample order (any int array) as the device array `cut` indexes"""
        return torch.as_tensor(np.ascontiguousarray(order, dtype=np.int64)).to(self.desc.device)

    def cut(self, order, first, n, size, mode=ZNORM, shifted=False, flip_x=False, out=None):
        """crops of samples order[first : first + n] (order None: samples first .. first + n - 1) -> (n, 1, cz, cy, cx)
        ((n, 1, cy, cx) for mode SUMZ_MINMAX)"""
        cz, cy, cx = [int(v) for v in size]
        if first < 0 or n < 0 or first + n > (self.n if order is None else int(order.numel())):
            raise L.HipExtensionError("CropTable.cut: samples [%d, %d) outside the epoch's order" % (first, first + n))
        shape = (n, 1, cy, cx) if mode == SUMZ_MINMAX else (n, 1, cz, cy, cx)
        if out is None:
            out = torch.empty(shape, dtype=torch.float32, device=self.desc.device)
        elif tuple(out.shape) != shape or out.dtype != torch.float32 or not out.is_contiguous():
            raise L.HipExtensionError("CropTable.cut: `out` must be a contiguous float32 %s tensor" % (shape,))
        L.check(L.lib().mi_crop_normalize_table(L.ptr(self.desc), L.ptr(self.owner), L.ptr(self.centres),
                                                L.ptr(self.shift if shifted else None), L.ptr(order), int(first), int(n),
                                                cz, cy, cx, int(mode), int(bool(flip_x)), L.ptr(out), L.stream()),
                "mi_crop_normalize_table")
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


def extract_3d_tomo(rec, tomo_coords, crop_size_x, crop_size_y):
    """datasets/tomo_pre_proj_angle_select_new3d_vol.py:109-115 for every pick: the min-max'ed patch of slice z,
    (n, 1, crop_y, crop_x)."""
    return _crop(rec, tomo_coords, (1, crop_size_y, crop_size_x), SUMZ_MINMAX).unsqueeze(1)


def subvol_mean_std(subvols):
    """...:238-239 `torch.mean(torch.stack(sub_vols_3d))`, `torch.std(...)` (unbiased) over all crops, on the device:
    returns two Python floats (one read-back per dataset, like the reference's two scalars)."""
    L.require_cuda(subvols, "subvols")
    x = subvols.contiguous().view(-1)
    n = x.numel()
    lib = L.lib()
    ws = L.workspace(lib.mi_vol_stats_workspace_bytes(1, n), x.device, "stats")
    st = torch.empty(4, dtype=torch.float64, device=x.device)
    L.check(lib.mi_vol_stats(L.ptr(x), 1, n, L.ptr(st), L.ptr(ws), ws.numel(), L.stream()), "mi_vol_stats")
    mean, std0 = float(st[0]), float(st[1])
    return mean, std0 * (n / (n - 1.0)) ** 0.5


def to_uint8_normalize(subvols, mean, std):
    """simsiam_test_hm_3d.py:45-51 `T.ToPILImage() -> T.ToTensor() -> T.Normalize(mean, std)` for a batch of min-max'ed
    crops (the 8-bit round trip truncates: floor(255 x) / 255)."""
    L.require_cuda(subvols, "subvols")
    x = subvols.contiguous()
    y = torch.empty_like(x)
    L.check(L.lib().mi_u8_roundtrip_normalize(L.ptr(x), L.ptr(y), x.numel(), float(mean), float(std), L.stream()),
            "mi_u8_roundtrip_normalize")
    return y


def cutup_centres(shape, size, stride, margin=(0, 0, 0)):
    """Centres (x, y, z) of the inner crop of every `cutup(v, size, stride)` window (utils/loader.py:124-132 as used at
    datasets/tomo_pre.py:104), window order (i, j, k) row-major; `margin` = voxels the chain's Crop removes per side.
    Returns (centres (n,3) int32, inner size)."""
    nb = [(int(shape[a]) - int(size[a])) // int(stride[a]) + 1 for a in range(3)]
    inner = [int(size[a]) - 2 * int(margin[a]) for a in range(3)]
    iz, iy, ix = np.meshgrid(np.arange(nb[0]), np.arange(nb[1]), np.arange(nb[2]), indexing="ij")
    o = [iz.ravel() * stride[0] + margin[0], iy.ravel() * stride[1] + margin[1], ix.ravel() * stride[2] + margin[2]]
    c = np.stack([o[2] + inner[2] // 2, o[1] + inner[1] // 2, o[0] + inner[0] // 2], 1).astype(np.int32)
    return c, tuple(inner)


def crop_znorm_rescale_znorm(v, tomo_coords, size, flip_x=False):
    """datasets/tomo_pre.py:57-60 (deterministic tail of the torchio chain) for every window: (n, 1, cz, cy, cx) =
    ZNormalization(RescaleIntensity(-3, 3)(ZNormalization(crop)))."""
    return _crop(v, tomo_coords, size, ZNORM_RESCALE_ZNORM, flip_x).unsqueeze(1)
