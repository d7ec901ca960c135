"""A dataset with the reference's batch contract ({'input', 'input_aug'}) over a synthetic tomogram:
sub-tomograms are cut and z-normalised on the GPU (datasets/subvols.py), the second view is the
crop shifted by <= 1 voxel and mirrored along x.  The reference's real datasets
(MRC I/O, random torchvision/torchio augmentations) are out of scope."""
import numpy as np
import torch

from ..synthetic import make_tomo
from . import subvols as S


class SyntheticMocoLoader:
    """Iterable of batches resident in HBM; `len()` = batches per epoch; `set_epoch` reshuffles
    like DistributedSampler.set_epoch (moco_main.py:155-156)."""

    def __init__(self, shape=(64, 256, 256), crop=32, n_crops=1024, batch_size=64, seed=317, device="cuda",
                 rank=0, world=1):
        vol, _ = make_tomo(shape, seed=seed + rank)
        self.vol = torch.as_tensor(vol).to(device)
        g = np.random.default_rng(seed + rank)
        z, h, w = shape
        hc = crop // 2
        self.centres = np.stack([g.integers(hc + 1, w - hc - 1, n_crops), g.integers(hc + 1, h - hc - 1, n_crops),
                                 g.integers(hc + 1, z - hc - 1, n_crops)], 1).astype(np.int32)
        self.shift = g.integers(-1, 2, (n_crops, 3)).astype(np.int32)
        self.crop, self.batch_size, self.seed, self.epoch = crop, batch_size, seed, 0
        self.table = S.CropTable([self.vol], np.zeros(n_crops, dtype=np.int32), self.centres, self.shift)

    def set_epoch(self, epoch):
        self.epoch = epoch

    def __len__(self):
        return len(self.centres) // self.batch_size

    def __iter__(self):
        # (the epoch's order goes to the device once; a batch is two launches of mi_crop_normalize_table)
        order = self.table.epoch_order(np.random.default_rng(self.seed + 1000 * self.epoch).permutation(len(self.centres)))
        c, B = (self.crop,) * 3, self.batch_size
        for b in range(len(self)):
            yield {"input": self.table.cut(order, b * B, B, c),
                   "input_aug": self.table.cut(order, b * B, B, c, shifted=True, flip_x=True)}
