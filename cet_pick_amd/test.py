"""`python -m cet_pick_amd.test semi --arch unet_4 --load_model ... --with_score` - the reference's cet_pick/test.py
(:65-97): the detector over every tomogram of the test split; per tomogram `{name}.txt` (x<TAB>z<TAB>y[<TAB>score]) and
`{name}_hm.mrc` under `--out_id`, and the reference's timing keys.  Tomograms: labelled synthetic volumes
(datasets/synthetic_datasets.py) or, with `--test_img_txt <file>` naming MRC files (one `name<TAB>path` per line after a
header), real reconstructions through the device loader (utils/loader.py `load_rec` + `preprocess`)."""
import os

import torch

from .datasets.synthetic_datasets import SyntheticDetectorDataset
from .detectors.detector_factory import detector_factory
from .opts import opts
from .utils.utils import AverageMeter, TextLog


LAST_STAGES = {}         # seconds of the last tomogram's stages outside BaseDetector.run's dict (tools/bench_infer_entry.py)


def _mrc_list(path, opt):
    import time
    from .utils import loader
    rows = [ln.split("\t") for ln in open(path).read().splitlines()[1:] if ln.strip()]
    for name, p in rows:
        t0 = time.time()
        rec = loader.preprocess(loader.load_rec(p, order=opt.order, compress=opt.compress), opt.gauss)
        torch.cuda.synchronize()
        LAST_STAGES["file_to_device"] = time.time() - t0
        yield {"input": rec[None].float(), "meta": {"name": [name], "zdim": int(rec.shape[0])}}


def test(opt):
    Dataset = SyntheticDetectorDataset
    opt = opts().update_dataset_info_and_set_heads(opt, Dataset)
    TextLog(opt).close()
    detector = detector_factory[opt.task](opt)
    path = os.path.join(opt.data_dir, opt.test_img_txt)
    loader = _mrc_list(path, opt) if os.path.exists(path) else Dataset(opt, "test")
    time_stats = ["tot_time", "load", "pre", "net", "dec"]
    avg_time_stats = {t: AverageMeter() for t in time_stats}
    n = 0
    for batch in loader:
        ret = detector.run(batch["input"], batch["meta"])
        LAST_STAGES.update(getattr(detector, "last_stages", {}))
        for t in avg_time_stats:
            avg_time_stats[t].update(ret[t])
        n += 1
        print("[{0}] ".format(n) + "".join("|{} {tm.val:.3f}s ({tm.avg:.3f}s) ".format(t, tm=avg_time_stats[t])
                                           for t in avg_time_stats))
    return {t: avg_time_stats[t].avg for t in avg_time_stats}


if __name__ == "__main__":
    test(opts().parse())
