"""Synthetic datasets with the batch contracts of the reference's datasets (which are outside the hot path: MRC file
lists, torchvision / torchio augmentation).  They run the hot-path stages the real datasets run - the DoG picker, the crop
kernels, the dataset normalisation - on synthetic tomograms (cet_pick_amd.synthetic.make_tomo), resident in HBM.

  SyntheticSimSiamDataset   datasets/tomo_pre_proj_angle_select_new3d_vol.py (load_data :160-241) +
                            particle_pre_3d_vol.py: {'input', 'input_aug'} of (B, 1, bbox, bbox) crops; the attributes
                            simsiam_test_hm_3d.py reads (sub_vols_3d, mean_subvols3d, std_subvols3d, names_all, coords)
  SyntheticDetectorDataset  datasets/tomo_moco.py + particle_moco.py: {'input', 'input_aug', 'hm', 'flip_prob'} crop
                            pairs for training; `.images` / `.names` for test.py
"""
import numpy as np
import torch

from ..synthetic import make_tomo
from ..utils import image as Im
from . import subvols as S


class SyntheticSimSiamDataset:
    """One or more synthetic tomograms -> DoG picks (`get_potential_coords_pyramid`, sigma = --dog) -> (3, bbox, bbox)
    crops summed over z and min-max'ed (`extract_subvols`) -> dataset mean / std (:238-239).  Iterating yields
    batches of two views: the normalised crop and its mirror image (the random torchvision augmentations of the
    reference's sample class are out of scope)."""
    num_classes = 256
    default_resolution = [24, 24]

    def __init__(self, opt, split, size, sigma1=(2.5, 5), shape=(24, 256, 256), n_tomos=2, device="cuda", rank=0, world=1,
                 max_per_tomo=512):
        self.opt, self.split, self.size = opt, split, tuple(int(s) for s in size)
        self.batch_size = max(1, int(getattr(opt, "batch_size", 8)))
        self.rank, self.world, self.epoch, self.seed = rank, world, 0, int(getattr(opt, "seed", 317))
        self.tomos, self.names, self.names_all, self.coords = {}, [], [], []
        crops = []
        b = self.size[1]
        for t in range(n_tomos):
            vol, _ = make_tomo(shape, seed=self.seed + t, margin_xy=min(40, shape[1] // 4), margin_z=min(12, shape[0] // 4))
            name = "synthetic_%d" % t
            rec = torch.as_tensor(vol).to(device)
            # (a literal two-slice tomogram has no picks under the reference's 10-slice z border: it is a parameter here)
            _, c = Im.get_potential_coords_pyramid(rec, sigmas=list(sigma1), border_z=min(10, shape[0] // 4))
            keep = ((c[:, 0] >= b // 2 + 1) & (c[:, 0] < shape[2] - b // 2 - 1) & (c[:, 1] >= b // 2 + 1) &
                    (c[:, 1] < shape[1] - b // 2 - 1) & (c[:, 2] >= 1) & (c[:, 2] < shape[0] - 1))
            c = c[keep][:max_per_tomo]
            if len(c) == 0:
                continue
            crops.append(S.extract_subvols(rec, c, self.size))
            self.tomos[name] = rec
            self.names.append(name)
            self.names_all += [name] * len(c)
            self.coords += [row for row in c]
        if not crops:
            raise RuntimeError("the DoG picker found no particle on the synthetic tomograms")
        self.sub_vols_3d = torch.cat(crops, 0)                                   # (n, 1, bbox, bbox) in [0, 1]
        self.mean_subvols3d, self.std_subvols3d = S.subvol_mean_std(self.sub_vols_3d)
        self.normed = (self.sub_vols_3d - self.mean_subvols3d) / self.std_subvols3d
        self.num_samples = self.sub_vols_3d.shape[0]

    def set_epoch(self, epoch):
        self.epoch = epoch

    def __len__(self):                      # batches per epoch and rank (drop_last, like the reference's loaders)
        return (self.num_samples // self.world) // self.batch_size

    def __iter__(self):
        order = np.random.default_rng(self.seed + 1000 * self.epoch).permutation(self.num_samples)
        order = order[self.rank::self.world]
        for i in range(len(self)):
            idx = torch.as_tensor(order[i * self.batch_size:(i + 1) * self.batch_size], device=self.normed.device)
            x = self.normed[idx]
            yield {"input": x, "input_aug": x.flip(-1).contiguous()}


def _splat(hm, cx, cy, cz, radius=2):
    """Gaussian label blob with peak 1 (the reference draws `draw_umich_gaussian_3d`; label drawing is dataset code)."""
    d, h, w = hm.shape
    z0, z1 = max(0, cz - 1), min(d, cz + 2)
    y0, y1 = max(0, cy - radius), min(h, cy + radius + 1)
    x0, x1 = max(0, cx - radius), min(w, cx + radius + 1)
    zz, yy, xx = np.ogrid[z0:z1, y0:y1, x0:x1]
    g = np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2 + 4.0 * (zz - cz) ** 2) / (2.0 * (radius / 1.5) ** 2)).astype(np.float32)
    hm[z0:z1, y0:y1, x0:x1] = np.maximum(hm[z0:z1, y0:y1, x0:x1], g)


class SyntheticDetectorDataset:
    """Labelled synthetic tomograms for the CenterNet-3D detector (task 'semi').  Training batches are pairs of
    (crop_d, crop, crop) crops - the second view mirrored along x (flip_prob <= 0.5) or y - with the semi-supervised
    label volume at half xy resolution: 1-peaked blobs on half of the true particles, 0 on a band of labelled
    background, -1 (unlabeled) elsewhere; every crop holds at least one labelled particle (the PU loss raises without
    positives, models/loss.py:275-276)."""
    num_classes = 1
    default_resolution = [64, 64]

    def __init__(self, opt, split, shape=(32, 256, 256), n_tomos=2, crop_d=6, crop=64, per_epoch=64, device="cuda", rank=0,
                 world=1):
        self.opt, self.split = opt, split
        self.batch_size = max(1, int(getattr(opt, "batch_size", 1)))
        self.rank, self.world, self.epoch, self.seed = rank, world, 0, int(getattr(opt, "seed", 317))
        self.crop_d, self.crop, self.per_epoch, self.device = crop_d, crop, per_epoch, device
        self.images, self.names, self.centres = [], [], []
        for t in range(n_tomos):
            vol, centres = make_tomo(shape, seed=self.seed + 50 + t, margin_xy=min(40, shape[1] // 4), margin_z=min(6, shape[0] // 4))
            v = vol.astype(np.float32)
            self.images.append((v - v.mean()) / v.std())
            self.names.append("synthetic_det_%d" % t)
            self.centres.append(centres)
        self.shape = shape

    def set_epoch(self, epoch):
        self.epoch = epoch

    def __len__(self):
        if self.split != "train":
            return len(self.images)
        return (self.per_epoch // self.world) // self.batch_size

    def _sample(self, rng):
        t = int(rng.integers(len(self.images)))
        vol, cen = self.images[t], self.centres[t]
        d, h, w = self.shape
        cd, c = self.crop_d, self.crop
        px, py, pz = cen[int(rng.integers(len(cen)))]
        z0 = int(np.clip(pz - cd // 2 + rng.integers(-1, 2), 0, d - cd))
        y0 = int(np.clip(py - c // 2 + rng.integers(-12, 13), 0, h - c))
        x0 = int(np.clip(px - c // 2 + rng.integers(-12, 13), 0, w - c))
        x = vol[z0:z0 + cd, y0:y0 + c, x0:x0 + c]
        hm = np.full((cd, c // 2, c // 2), -1.0, np.float32)
        hm[:, :4, :] = 0.0                                              # a labelled-background band
        n_pos = 0
        for k, (qx, qy, qz) in enumerate(cen):
            if z0 <= qz < z0 + cd and y0 <= qy < y0 + c and x0 <= qx < x0 + c and (k % 2 == 0 or (qx, qy, qz) == (px, py, pz)):
                _splat(hm, (qx - x0) // 2, (qy - y0) // 2, qz - z0)
                n_pos += 1
        return x, hm, n_pos

    def __iter__(self):
        if self.split != "train":
            for img, name in zip(self.images, self.names):
                yield {"input": torch.as_tensor(img)[None], "meta": {"name": [name], "zdim": img.shape[0]}}
            return
        rng = np.random.default_rng(self.seed + 1000 * self.epoch + self.rank)
        for _ in range(len(self)):
            xs, hms = [], []
            while len(xs) < self.batch_size:
                x, hm, n_pos = self._sample(rng)
                if n_pos:
                    xs.append(x); hms.append(hm)
            flip_prob = float(rng.random())
            x = torch.as_tensor(np.stack(xs)).to(self.device)
            aug = x.flip(-2 if flip_prob > 0.5 else -1).contiguous()
            yield {"input": x, "input_aug": aug, "hm": torch.as_tensor(np.stack(hms))[:, None].to(self.device),
                   "flip_prob": flip_prob, "meta": {}}
