"""File-backed datasets for the train mains: the reference's `--train_img_txt` list of MRC reconstructions
(datasets/tomo_pre_proj_angle_select_new3d_vol.py:29-93,160-241 `load_data`; datasets/tomo_pre.py:24-70,90-128;
datasets/particle_pre_3d_vol.py:70-85 for the batch dictionary) through the device pipeline of this build:

    list (tab-separated, header `image_name<TAB>rec_path`)  ->  utils.loader.load_tomos_from_list (`mi_rec_reorder`, `mi_zscore`,
    `mi_zscore_quantize_minmax`, optional Gaussian denoise)  ->  DoG picks (`mi_dog_pick`, sigma = --dog)  ->  crops
    (`mi_crop_normalize`)  ->  the batch contract the synthetic datasets have ({'input', 'input_aug'}).

Semantics kept from the reference: the list format, `--order` / `--compress` / `--gauss`, the picker and its sigmas, the
crop geometry and the border rule of `load_data` (:196), the dataset mean / std (:238-239), the attributes
simsiam_test_hm_3d.py reads.  Out of scope: the random torchvision / torchio augmentations - the second
view is the mirrored crop (2-D) / the crop shifted by <= 1 voxel and mirrored (3-D), as in the synthetic datasets.
"""
import os

import numpy as np
import torch

from ..utils import image as Im
from ..utils import loader as Ld
from . import subvols as S
from .synthetic_datasets import SyntheticSimSiamDataset
from .synthetic_moco import SyntheticMocoLoader


def read_image_list(path):
    """The reference reads the list with pandas.read_csv(sep='\\t') and uses the columns `image_name` and `rec_path`
    (:161-163).  Relative paths are taken relative to the list's directory.  -> [(name, path)]"""
    if not os.path.isfile(path):
        raise FileNotFoundError("image list %s not found (--train_img_txt under <cwd>/data; --dataset synthetic trains on a "
                                "synthetic tomogram instead)" % path)
    with open(path) as f:
        lines = [ln.rstrip("\n").rstrip("\r") for ln in f if ln.strip()]
    header = lines[0].split("\t")
    if "image_name" not in header or "rec_path" not in header:
        raise ValueError("%s: the header needs the tab-separated columns image_name and rec_path, got %s" % (path, header))
    i_name, i_path = header.index("image_name"), header.index("rec_path")
    base = os.path.dirname(os.path.abspath(path))
    out = []
    for ln in lines[1:]:
        cols = ln.split("\t")
        p = cols[i_path]
        out.append((cols[i_name], p if os.path.isabs(p) else os.path.join(base, p)))
    if not out:
        raise ValueError("%s lists no tomogram" % path)
    return out


def use_files(opt, split="train"):
    """True when the run reads tomograms from files: any `--dataset` other than 'synthetic' whose image list exists.  A
    missing list is announced (one line on stdout) and the synthetic tomogram takes its place - the plumbing configurations
    of BASELINE.json run without a data directory."""
    if getattr(opt, "dataset", "synthetic") == "synthetic":
        return False
    path = os.path.join(opt.data_dir, opt.train_img_txt if split == "train" else opt.test_img_txt)
    if os.path.isfile(path):
        return True
    print("[cet_pick_amd] no image list at %s: --dataset %s runs on the synthetic tomogram" % (path, opt.dataset))
    return False


def load_listed_tomos(opt, split="train"):
    """name -> (Z', H, W) device tensor in [0, 1], as `load_tomos_from_list` (loader.py:165-173) returns them."""
    txt = opt.train_img_txt if split == "train" else opt.test_img_txt
    items = read_image_list(os.path.join(opt.data_dir, txt))
    return Ld.load_tomos_from_list([n for n, _ in items], [p for _, p in items], order=opt.order, compress=opt.compress,
                                   denoise=opt.gauss)


class TomoFileSimSiamDataset(SyntheticSimSiamDataset):
    """`TOMOPreProjAngleSelect3DVol` (:25-241) on the device: listed tomograms -> DoG picks -> the border rule of :196
    (x and y at least crop // 1.8 from the edges; z inside the volume) -> (sz, bbox, bbox) crops summed over z and min-max'ed
    (`extract_subvols`) -> dataset mean / std.  Same attributes and batches as SyntheticSimSiamDataset."""

    def __init__(self, opt, split, size, sigma1=(2.5, 5), device="cuda", rank=0, world=1):
        self.opt, self.split, self.size = opt, split, tuple(int(s) for s in size)
        self.batch_size = max(1, int(getattr(opt, "batch_size", 8)))
        self.rank, self.world, self.epoch, self.seed = rank, world, 0, int(getattr(opt, "seed", 317))
        self.tomos, self.names, self.names_all, self.coords = {}, [], [], []
        crops = []
        cx, cy = self.size[1], self.size[2]
        for name, rec in load_listed_tomos(opt, split).items():
            d, h, w = rec.shape
            _, c = Im.get_potential_coords_pyramid(rec, sigmas=list(sigma1))
            mx, my = cx // 1.8, cy // 1.8                                          # :196
            keep = (c[:, 0] > mx) & (c[:, 0] < w - mx) & (c[:, 1] >= my) & (c[:, 1] <= h - my) & (c[:, 2] >= 1) & (c[:, 2] < d - 1)
            c = c[keep]
            self.tomos[name] = rec
            self.names.append(name)
            if len(c) == 0:
                continue
            crops.append(S.extract_subvols(rec, c, self.size))
            self.names_all += [name] * len(c)
            self.coords += [row for row in c]
        if not crops:
            raise RuntimeError("the DoG picker found no particle on the listed tomograms (sigma %s)" % (list(sigma1),))
        self.sub_vols_3d = torch.cat(crops, 0)
        self.mean_subvols3d, self.std_subvols3d = S.subvol_mean_std(self.sub_vols_3d)
        self.normed = (self.sub_vols_3d - self.mean_subvols3d) / self.std_subvols3d
        self.num_samples = self.sub_vols_3d.shape[0]
        print("Loaded {} {} samples".format(split, self.num_samples))


class TomoFileMocoLoader(SyntheticMocoLoader):
    """3-D sub-tomogram pairs for moco_main from listed tomograms: crop centres = the DoG picks that keep a whole crop^3 box
    inside the volume; views = the z-normalised crop and the crop shifted by <= 1 voxel, mirrored along x (what
    SyntheticMocoLoader serves).  A batch may mix tomograms: every tomogram's share is cut in one launch."""

    def __init__(self, opt, crop=32, device="cuda", rank=0, world=1, split="train"):
        self.vols, cents, owner = [], [], []
        hc = crop // 2
        for name, rec in load_listed_tomos(opt, split).items():
            d, h, w = rec.shape
            _, c = Im.get_potential_coords_pyramid(rec, sigmas=list(opt.dog), border_z=min(10, max(d // 4, 1)))
            keep = ((c[:, 0] >= hc + 1) & (c[:, 0] < w - hc - 1) & (c[:, 1] >= hc + 1) & (c[:, 1] < h - hc - 1) &
                    (c[:, 2] >= hc + 1) & (c[:, 2] < d - hc - 1))
            c = c[keep]
            if len(c):
                owner += [len(self.vols)] * len(c)
                self.vols.append(rec)
                cents.append(c.astype(np.int32))
        if not cents:
            raise RuntimeError("the DoG picker found no crop centre with a whole %d^3 box on the listed tomograms" % crop)
        self.centres = np.concatenate(cents, 0)
        self.owner = np.asarray(owner, dtype=np.int64)
        self.seed = int(getattr(opt, "seed", 317))
        g = np.random.default_rng(self.seed + rank)
        self.shift = g.integers(-1, 2, self.centres.shape).astype(np.int32)
        self.crop, self.batch_size, self.epoch = crop, int(opt.batch_size), 0
        self.rank, self.world = rank, world
        # the bookkeeping of every batch, on the device once: tomogram descriptors, owner / centre / shift per sample
        self.table = S.CropTable(self.vols, self.owner, self.centres, self.shift)
        self.prefetch = os.environ.get("CETPICK_LOADER_PREFETCH", "1") != "0"
        self._stream = None
        print("Loaded {} {} samples".format(split, len(self.centres)))

    def __len__(self):
        return (len(self.centres) // self.world) // self.batch_size

    def _cut(self, idx, shifted):
        """one view of the samples `idx` with the per-batch bookkeeping on the HOST (rounds 3-4: index arithmetic in numpy, an
        index upload and a scatter per tomogram); kept as the reference form of what `__iter__` serves
        (tests/test_entry_points_gpu.py::test_device_side_loader_serves_the_crops_of_the_host_side_one) - the batches themselves come from the device-side table"""
        c = (self.crop,) * 3
        out = torch.empty((len(idx), 1) + c, dtype=torch.float32, device=self.vols[0].device)
        for v in np.unique(self.owner[idx]):
            sel = np.nonzero(self.owner[idx] == v)[0]
            cen = self.centres[idx[sel]] + (self.shift[idx[sel]] if shifted else 0)
            out[torch.as_tensor(sel, device=out.device)] = S.crop_znorm(self.vols[v], cen, c, flip_x=shifted)
        return out

    def epoch_order(self):
        """this rank's sample order of the epoch (DistributedSampler semantics: one seeded permutation, rank-strided)"""
        return np.random.default_rng(self.seed + 1000 * self.epoch).permutation(len(self.centres))[self.rank::self.world]

    def __iter__(self):
        # one upload per EPOCH (the permutation); a batch is two launches that index it - no numpy slice, no np.unique, no
        # host-to-device copy, no scatter per tomogram (moco_main.py:122-156's DataLoader, minus its workers)
        order = self.table.epoch_order(self.epoch_order())
        c, B = (self.crop,) * 3, self.batch_size
        dev = self.vols[0].device
        if dev.type != "cuda" or not self.prefetch:
            for b in range(len(self)):
                yield {"input": self.table.cut(order, b * B, B, c),
                       "input_aug": self.table.cut(order, b * B, B, c, shifted=True, flip_x=True)}
            return
        # One batch ahead on a stream of the loader's own (what the reference's DataLoader workers are for): the two crop launches of
        # batch b + 1 run next to training step b instead of in front of step b + 1.  The consumer's stream waits for the batch's
        # event; the tensors are handed over with record_stream (the allocator must not recycle them for the loader's stream early).
        if self._stream is None:
            self._stream = torch.cuda.Stream(device=dev)
        side = self._stream

        def cut(b):
            if b == 0:
                side.wait_stream(torch.cuda.current_stream(dev))    # (the epoch's order upload; the table has been there since __init__)
            with torch.cuda.stream(side):
                batch = {"input": self.table.cut(order, b * B, B, c),
                         "input_aug": self.table.cut(order, b * B, B, c, shifted=True, flip_x=True)}
                ev = torch.cuda.Event()
                ev.record(side)
            return batch, ev

        nxt = cut(0) if len(self) else None
        for b in range(len(self)):
            batch, ev = nxt
            nxt = cut(b + 1) if b + 1 < len(self) else None
            cur = torch.cuda.current_stream(dev)
            cur.wait_event(ev)
            for t in batch.values():
                t.record_stream(cur)
            yield batch
