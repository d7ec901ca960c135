"""`python -m cet_pick_amd.simsiam_test_hm_3d simsiam3d --arch simsiam2d_18 --load_model ...` - the reference's
cet_pick/simsiam_test_hm_3d.py (:136-195): exploration inference.  Every picked crop goes through the 8-bit round trip +
`Normalize(dataset mean, std)` of `PrefetchDatasetProj` (:45-51, on the device: datasets/subvols.py `to_uint8_normalize`),
the encoder's `forward_test`, and `all_output_info.npz` = {proj, pred, name, coords, subvol} is written under save_dir
for the downstream 2-D plots (:190-195)."""
import os

import numpy as np
import torch

from .datasets import subvols as S
from .datasets.synthetic_datasets import SyntheticSimSiamDataset
from .models.model import create_model, load_model
from .opts import opts
from .utils.utils import TextLog


LAST_STAGES = {}         # seconds of the last call's stages (tools/bench_infer_entry.py)


def test(opt):
    import time
    t_start = time.time()
    Dataset = SyntheticSimSiamDataset
    opt = opts().update_dataset_info_and_set_heads(opt, Dataset)
    TextLog(opt).close()
    if opt.gpus[0] < 0:
        raise RuntimeError("the MI355X path has no CPU mode (--gpus -1)")
    opt.device = torch.device("cuda", opt.gpus[0])
    model = create_model(opt.arch, opt.heads, opt.head_conv)
    if opt.load_model != "":
        model = load_model(model, opt.load_model)
    model = model.to(opt.device)
    model.eval()
    from .datasets.tomo_files import use_files
    if use_files(opt, "test"):
        from .datasets.tomo_files import TomoFileSimSiamDataset as Dataset
    t_model = time.time()
    dataset = Dataset(opt, "test", (3, opt.bbox, opt.bbox), sigma1=opt.dog, device=opt.device)
    torch.cuda.synchronize()
    t_data = time.time()
    normed = S.to_uint8_normalize(dataset.sub_vols_3d, dataset.mean_subvols3d, dataset.std_subvols3d)
    all_proj, all_pred = [], []
    with torch.no_grad():
        for i in range(0, normed.shape[0], 256):                        # batch_size=256 (:150)
            ret = model.forward_test(normed[i:i + 256].contiguous())
            all_proj.append(ret["proj"].detach())                       # (stay on the device: one copy to the host at the end, not three
            all_pred.append(ret["pred"].detach())                       #  synchronising copies per batch as :163-176 makes)
    proj, pred = torch.cat(all_proj, 0).cpu().numpy(), torch.cat(all_pred, 0).cpu().numpy()
    subvol = normed.cpu().numpy()
    t_net = time.time()
    out_file = os.path.join(opt.save_dir, "all_output_info.npz")
    np.savez(out_file, proj=proj, pred=pred, name=np.asarray(dataset.names_all), coords=np.asarray(dataset.coords), subvol=subvol)
    LAST_STAGES.update({"model": t_model - t_start, "load_pick_crop": t_data - t_model, "net": t_net - t_data, "save": time.time() - t_net})
    print("opt.save_dir", opt.save_dir)
    return out_file


if __name__ == "__main__":
    test(opts().parse())
