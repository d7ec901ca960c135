"""The reference's entry scripts besides moco_main, end to end on the GPU: simsiam_main.py,
simsiam_test_hm_3d.py, main.py, test.py - flags -> model factory -> trainer / detector -> the reference's log line,
checkpoint names and output files, on the synthetic datasets."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_simsiam_main_then_exploration_inference(tmp_path, monkeypatch):
    from cet_pick_amd import simsiam_main, simsiam_test_hm_3d
    from cet_pick_amd.opts import opts
    monkeypatch.chdir(tmp_path)
    args = ["simsiam3d", "--arch", "simsiam2d_18", "--dataset", "simsiam3d", "--bbox", "36", "--batch_size", "8",
            "--num_epochs", "2", "--num_iters", "6", "--lr", "0.01", "--val_intervals", "2", "--exp_id", "s", "--debug", "0",
            "--dog", "2.5,5"]
    simsiam_main.main(opts().parse(args))
    save_dir = os.path.join(str(tmp_path), "exp", "simsiam3d", "s")
    lines = open(os.path.join(save_dir, "log.txt")).read().strip().split("\n")
    assert len(lines) == 2 and lines[0].startswith("epoch: 1 |loss ")
    for key in ("cosine_loss", "output_std", "time"):
        assert key in lines[0]
    loss = [float(l.split("|")[1].split()[1]) for l in lines]
    assert all(np.isfinite(loss)) and all(-1.0 <= v <= 0.0 for v in loss)            # negative cosine similarity
    assert os.path.exists(os.path.join(save_dir, "model_last_contrastive.pth"))          # epoch 1
    ck = torch.load(os.path.join(save_dir, "model_last.pth"))                            # epoch 2 (val interval)
    assert set(ck) == {"epoch", "state_dict", "optimizer"} and ck["epoch"] == 2
    # exploration inference on the trained encoder: all_output_info.npz (simsiam_test_hm_3d.py:190-195)
    out = simsiam_test_hm_3d.test(opts().parse(["simsiam3d", "--arch", "simsiam2d_18", "--dataset", "simsiam3d", "--bbox", "36",
                                                "--exp_id", "s", "--debug", "0", "--dog", "2.5,5",
                                                "--load_model", os.path.join(save_dir, "model_last.pth")]))
    z = np.load(out)
    assert set(z.files) == {"proj", "pred", "name", "coords", "subvol"}
    n = z["proj"].shape[0]
    assert n > 16 and z["proj"].shape == (n, 128) and z["pred"].shape == (n, 128)
    assert z["coords"].shape == (n, 3) and z["name"].shape == (n,) and z["subvol"].shape == (n, 1, 36, 36)
    assert np.isfinite(z["proj"]).all() and float(np.std(z["proj"], axis=0).mean()) > 0
    # the stored sub-volumes are the 8-bit round trip + Normalize of the min-max'ed crops: 256 levels at most
    lv = np.unique(np.round(z["subvol"][0] * 1e4) / 1e4)
    assert len(lv) <= 256


def test_detector_main_then_test(tmp_path, monkeypatch):
    from cet_pick_amd import main as det_main, test as det_test
    from cet_pick_amd.opts import opts
    from cet_pick_amd.utils import mrc
    monkeypatch.chdir(tmp_path)
    args = ["semi", "--arch", "unet_4", "--contrastive", "--batch_size", "2", "--num_epochs", "2", "--num_iters", "3",
            "--lr", "0.001", "--lr_step", "1", "--val_intervals", "2", "--exp_id", "d", "--debug", "0"]
    det_main.main(opts().parse(args))
    save_dir = os.path.join(str(tmp_path), "exp", "semi", "d")
    lines = open(os.path.join(save_dir, "log.txt")).read().strip().split("\n")
    assert len(lines) == 2 and lines[0].startswith("epoch: 1 |loss ")
    for key in ("hm_loss", "cr_loss", "consis_loss", "time"):
        assert key in lines[0]
    assert lines[1].count("hm_loss") == 2                                               # train + val columns
    for f in ("model_last_contrastive.pth", "model_1.pth", "model_last.pth", "model_best_contrastive.pth"):
        assert os.path.exists(os.path.join(save_dir, f)), f
    assert set(torch.load(os.path.join(save_dir, "model_best_contrastive.pth"))) == {"epoch", "state_dict"}
    # test.py on the trained detector: one coordinate file and one heat-map per tomogram
    times = det_test.test(opts().parse(["semi", "--arch", "unet_4", "--exp_id", "d", "--debug", "0", "--with_score", "--K", "100",
                                        "--cutoff_z", "1", "--out_thresh", "0.0", "--out_id", "picks",
                                        "--load_model", os.path.join(save_dir, "model_last.pth")]))
    assert set(times) == {"tot_time", "load", "pre", "net", "dec"}
    out = os.path.join(save_dir, "picks")
    for name in ("synthetic_det_0", "synthetic_det_1"):
        hm, _ = mrc.parse_mrc(os.path.join(out, name + "_hm.mrc"))
        assert hm.shape == (128, 32, 128) and np.isfinite(hm).all()                 # (H', D, W') as tomo_det.py:60 writes it
        rows = [ln.split("\t") for ln in open(os.path.join(out, name + ".txt")).read().splitlines()]
        assert all(len(r) == 4 for r in rows)
