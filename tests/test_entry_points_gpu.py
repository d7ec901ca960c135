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


def _write_listed_tomograms(tmp_path, shape=(48, 160, 176), n=2):
    """Two synthetic tomograms as MRC files + the reference's tab-separated image list under <cwd>/data."""
    from cet_pick_amd.synthetic import make_tomo
    from cet_pick_amd.utils import mrc
    data = tmp_path / "data"
    data.mkdir()
    lines = ["image_name\trec_path"]
    vols = {}
    for i in range(n):
        vol, _ = make_tomo(shape, seed=500 + i, margin_xy=40, margin_z=12)
        name = "tomo%d" % i
        mrc.write(str(data / (name + ".rec")), vol)
        lines.append("%s\t%s.rec" % (name, name))                  # relative to the list
        vols[name] = vol
    (data / "train_images.txt").write_text("\n".join(lines) + "\n")
    (data / "test_images.txt").write_text("\n".join(lines) + "\n")
    return vols


def test_moco_main_trains_on_mrc_files(tmp_path, monkeypatch):
    """`--train_img_txt` list of MRC files -> device load_rec / preprocess -> DoG picks -> crop kernel ->
    the batch contract of the step engine (datasets/tomo_pre_proj_angle_select_new3d_vol.py:29-93,181-239 semantics)."""
    from cet_pick_amd import moco_main
    from cet_pick_amd.datasets.tomo_files import TomoFileMocoLoader, read_image_list
    from cet_pick_amd.opts import opts
    monkeypatch.chdir(tmp_path)
    vols = _write_listed_tomograms(tmp_path)
    args = ["moco", "--arch", "moco3d_18", "--dataset", "simsiam3d", "--order", "zxy", "--batch_size", "8", "--num_epochs", "1",
            "--lr", "0.01", "--exp_id", "f", "--debug", "0", "--dog", "2.5,5", "--num_iters", "4"]
    o = opts().parse(args)
    assert [n for n, _ in read_image_list(os.path.join(o.data_dir, o.train_img_txt))] == ["tomo0", "tomo1"]
    loader = TomoFileMocoLoader(o, crop=32, device="cuda")
    assert len(loader.vols) == 2 and tuple(loader.vols[0].shape) == (48, 160, 176) and len(loader) >= 2
    c = loader.centres
    assert (c[:, 0] >= 17).all() and (c[:, 0] < 176 - 17).all() and (c[:, 2] >= 17).all() and (c[:, 2] < 48 - 17).all()
    batch = next(iter(loader))
    assert batch["input"].shape == batch["input_aug"].shape == (8, 1, 32, 32, 32)
    x = batch["input"]
    assert abs(float(x.mean())) < 1e-3 and abs(float(x.flatten(1).std(1).mean()) - 1.0) < 1e-3           # z-normalised crops
    # the loaded volume is the reference's preprocess() of the file: values on the 8-bit grid in [0, 1]
    v = loader.vols[0]
    assert float(v.min()) == 0.0 and float(v.max()) == 1.0
    moco_main.main(o)
    save_dir = os.path.join(str(tmp_path), "exp", "moco", "f")
    line = open(os.path.join(save_dir, "log.txt")).read().strip().split("\n")[-1]
    assert line.startswith("epoch: 1 |loss ") and np.isfinite(float(line.split("|")[1].split()[1]))
    assert os.path.exists(os.path.join(save_dir, "model_last_contrastive.pth"))


def test_device_side_loader_serves_the_crops_of_the_host_side_one(tmp_path, monkeypatch):
    """the MoCo loader's per-batch bookkeeping lives on the device (mi_crop_normalize_table: the epoch's
    order uploaded once, a batch = two launches that index the per-sample tables).  One epoch of it - batches that mix the
    two tomograms, both views, two ranks' strided shares - must be bit for bit what the host-side form (`_cut`: numpy index
    slices, an upload and a scatter per tomogram and view) serves."""
    from cet_pick_amd.datasets.tomo_files import TomoFileMocoLoader
    from cet_pick_amd.datasets.synthetic_moco import SyntheticMocoLoader
    from cet_pick_amd.datasets import subvols as S
    from cet_pick_amd.opts import opts
    monkeypatch.chdir(tmp_path)
    _write_listed_tomograms(tmp_path)
    o = opts().parse(["moco", "--arch", "moco3d_18", "--dataset", "simsiam3d", "--order", "zxy", "--batch_size", "16",
                      "--exp_id", "g", "--debug", "0", "--dog", "2.5,5"])
    for rank, world in ((0, 1), (1, 2)):
        loader = TomoFileMocoLoader(o, crop=32, device="cuda", rank=rank, world=world)
        assert len(set(loader.owner.tolist())) == 2
        loader.set_epoch(3)
        order = loader.epoch_order()
        n = 0
        for b, batch in enumerate(loader):
            idx = order[b * 16:(b + 1) * 16]
            assert torch.equal(batch["input"], loader._cut(idx, False)), (rank, b)
            assert torch.equal(batch["input_aug"], loader._cut(idx, True)), (rank, b)
            n += 1
        assert n == len(loader) >= 2
    # the synthetic loader takes the same path
    sl = SyntheticMocoLoader(shape=(40, 96, 96), crop=32, n_crops=64, batch_size=16, seed=5)
    sl.set_epoch(2)
    order = np.random.default_rng(5 + 2000).permutation(64)
    for b, batch in enumerate(sl):
        idx = order[b * 16:(b + 1) * 16]
        assert torch.equal(batch["input"], S.crop_znorm(sl.vol, sl.centres[idx], (32,) * 3))
        assert torch.equal(batch["input_aug"], S.crop_znorm(sl.vol, sl.centres[idx] + sl.shift[idx], (32,) * 3, flip_x=True))


def test_simsiam_main_and_exploration_on_mrc_files(tmp_path, monkeypatch):
    from cet_pick_amd import simsiam_main, simsiam_test_hm_3d
    from cet_pick_amd.opts import opts
    monkeypatch.chdir(tmp_path)
    _write_listed_tomograms(tmp_path, shape=(40, 200, 200))
    common = ["simsiam3d", "--arch", "simsiam2d_18", "--dataset", "simsiam3d", "--order", "zxy", "--bbox", "36", "--exp_id", "sf",
              "--debug", "0", "--dog", "2.5,5"]
    simsiam_main.main(opts().parse(common + ["--batch_size", "8", "--num_epochs", "1", "--num_iters", "4", "--lr", "0.01"]))
    save_dir = os.path.join(str(tmp_path), "exp", "simsiam3d", "sf")
    assert open(os.path.join(save_dir, "log.txt")).read().startswith("epoch: 1 |loss ")
    out = simsiam_test_hm_3d.test(opts().parse(common + ["--load_model", os.path.join(save_dir, "model_last_contrastive.pth")]))
    z = np.load(out)
    n = z["proj"].shape[0]
    assert n > 8 and set(np.unique(z["name"])) <= {"tomo0", "tomo1"} and z["subvol"].shape == (n, 1, 36, 36)
    # every pick respects the border rule of load_data (:196): crop // 1.8 = 20 voxels from the x / y edges
    assert (z["coords"][:, 0] > 20).all() and (z["coords"][:, 0] < 200 - 20).all()
