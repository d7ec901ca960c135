"""The inference entry points as a user runs them, on the C3 volume written as an MRC file:

  * `python -m cet_pick_amd.test semi --arch unet_4 --K 900 --with_score` (reference cet_pick/test.py:65-97 ->
    detectors/base_detector.py:62-106 `run` -> detectors/tomo_det.py:23-95): file -> device loader -> U-Net forward -> fused
    sigmoid + NMS + top-K -> post_process -> `{name}.txt` + `{name}_hm.mrc`; wall time per tomogram with the reference's own
    `load / pre / net / dec / tot` split plus `post` and `save`;
  * `python -m cet_pick_amd.simsiam_test_hm_3d simsiam3d --arch simsiam2d_18` (simsiam_test_hm_3d.py:136-195): file -> device loader ->
    DoG picks -> crops -> encoder `forward_test` -> `all_output_info.npz`.

    python tools/bench_infer_entry.py [--small]
"""
import json
import os
import shutil
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch


def run(small=False, repeats=3):
    from cet_pick_amd import test as det_test, simsiam_test_hm_3d
    from cet_pick_amd.models.model import create_model, save_model
    from cet_pick_amd.opts import opts
    from cet_pick_amd.synthetic import make_tomo, seeded_state_dict
    from cet_pick_amd.utils import mrc
    shape = (32, 128, 128) if small else (256, 512, 512)
    cwd = os.getcwd()
    tmp = tempfile.mkdtemp(prefix="cetpick_infer_entry_")
    out = {}
    try:
        os.chdir(tmp)
        os.makedirs("data")
        vol, _ = make_tomo(shape, seed=317)
        mrc.write(os.path.join("data", "c3.rec"), vol)
        for f in ("test_images.txt", "train_images.txt"):
            with open(os.path.join("data", f), "w") as fh:
                fh.write("image_name\trec_path\nc3\t%s\n" % os.path.join(tmp, "data", "c3.rec"))
        # ---- test.py: the CenterNet-3D detector ----
        heads = {"hm": 1, "proj": 32}
        net = create_model("unet_4", heads, 32)
        net.load_state_dict(seeded_state_dict(net, seed=321))
        ck = os.path.join(tmp, "unet4.pth")
        save_model(ck, 1, net)
        args = ["semi", "--arch", "unet_4", "--exp_id", "bench_infer", "--debug", "0", "--with_score", "--K", "900", "--order", "zxy",
                "--out_thresh", "0.0", "--out_id", "picks", "--load_model", ck]
        det_test.test(opts().parse(args))                                # warm-up: weight images, workspaces, allocator
        torch.cuda.synchronize()
        best = None
        for _ in range(repeats):
            t0 = time.perf_counter()
            times = det_test.test(opts().parse(args))
            torch.cuda.synchronize()
            wall = time.perf_counter() - t0
            if best is None or wall < best[0]:
                best = (wall, times, dict(det_test.LAST_STAGES))
        wall, times, stages = best
        n_vox = int(np.prod(shape))
        rec = {"workload": "cet_pick_amd.test semi --arch unet_4 --K 900 --with_score on one %s tomogram read from an MRC file "
                           "(BASELINE configs[2] volume)" % "x".join(map(str, shape)),
               "wall_ms_per_tomogram": wall * 1e3, "input_voxels_per_sec": n_vox / wall,
               "tot_ms": times["tot_time"] * 1e3, "load_ms": times["load"] * 1e3, "pre_ms": times["pre"] * 1e3,
               "net_ms": times["net"] * 1e3, "dec_ms": times["dec"] * 1e3,
               "post_ms": stages.get("post", 0.0) * 1e3, "save_ms": stages.get("save", 0.0) * 1e3,
               "file_to_device_ms": stages.get("file_to_device", 0.0) * 1e3,
               "net_plus_dec_over_tot": (times["net"] + times["dec"]) / times["tot_time"],
               "stages": "tot = the reference's BaseDetector.run dict (base_detector.py:62-106): host-to-device + net + dec + post + save; "
                         "file_to_device = MRC read + load_rec + preprocess in front of it (test.py's loader); wall = the whole "
                         "test() call incl. model creation and checkpoint load, best of %d" % repeats}
        hm, _ = mrc.parse_mrc(os.path.join(opts().parse(args).save_dir, "picks", "c3_hm.mrc"))
        rec["hm_mrc_shape"] = list(hm.shape)
        rec["detections_written"] = sum(1 for _ in open(os.path.join(opts().parse(args).save_dir, "picks", "c3.txt")))
        out["test_py_detector"] = rec
        # ---- simsiam_test_hm_3d.py: exploration inference ----
        enc = create_model("simsiam2d_18", {"proj": 128, "pred": 128}, 128)
        enc.load_state_dict(seeded_state_dict(enc, seed=318))
        ck2 = os.path.join(tmp, "simsiam2d.pth")
        save_model(ck2, 1, enc)
        args2 = ["simsiam3d", "--arch", "simsiam2d_18", "--dataset", "simsiam3d", "--order", "zxy", "--bbox", "36", "--exp_id", "bench_infer2",
                 "--debug", "0", "--dog", "3,5", "--load_model", ck2]
        simsiam_test_hm_3d.test(opts().parse(args2))
        torch.cuda.synchronize()
        best = None
        for _ in range(repeats):
            t0 = time.perf_counter()
            f = simsiam_test_hm_3d.test(opts().parse(args2))
            torch.cuda.synchronize()
            wall = time.perf_counter() - t0
            if best is None or wall < best[0]:
                best = (wall, dict(simsiam_test_hm_3d.LAST_STAGES))
        z = np.load(f)
        out["simsiam_test_hm_3d"] = {
            "workload": "cet_pick_amd.simsiam_test_hm_3d simsiam3d --arch simsiam2d_18 --bbox 36 --dog 3,5 on the same file",
            "wall_ms_per_tomogram": best[0] * 1e3, "input_voxels_per_sec": n_vox / best[0], "picks_embedded": int(z["proj"].shape[0]),
            **{k + "_ms": v * 1e3 for k, v in best[1].items()}}
        return out
    finally:
        os.chdir(cwd)
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    print(json.dumps(run(small="--small" in sys.argv)))
