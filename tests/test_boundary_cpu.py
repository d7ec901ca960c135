"""Host-side drop-in surface that needs no GPU: flags, lr schedule, checkpoint layout, factory."""
import json
import os

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def test_opts_defaults_and_derived_fields(tmp_path, monkeypatch):
    from cet_pick_amd.opts import opts
    monkeypatch.chdir(tmp_path)
    o = opts().parse(["moco", "--arch", "moco3d_18", "--batch_size", "64", "--gpus", "0,1", "--lr_step", "90,120",
                      "--debug", "0"])
    assert o.task == "moco" and o.arch == "moco3d_18" and o.gpus == [0, 1] and o.gpus_str == "0,1"
    assert o.lr_step == [90, 120] and o.seed == 317 and o.K == 200 and o.nms == 3 and o.dog == [2.5, 5]
    assert o.dist_backend == "nccl" and o.dist_url == "env://" and o.chunk_sizes == [32, 32]
    assert o.save_dir == os.path.join(str(tmp_path), "exp", "moco", "default")
    assert o.fix_res is True and o.cutoff_z == 10 and o.out_thresh == 0.25 and o.order == "xzy"
    o2 = opts().parse(["simsiam3d", "--dog", "3,5", "--resume"])
    assert o2.head_conv == 128 and o2.dog == [3.0, 5.0]
    assert o2.load_model.endswith(os.path.join("exp", "simsiam3d", "default", "model_last.pth"))
    assert opts().parse(["semi", "--gpus", "-1"]).gpus == [-1]
    o3 = opts().parse(["semi", "--warm", "--cosine"])       # the reference raises NameError here
    assert 0 < o3.warmup_to <= o3.lr
    o4 = opts().init(["moco"])
    assert o4.heads == {"proj": 256, "pred": 256} and o4.input_h == 32 and o4.dataset == "moco"
    o5 = opts().init(["semi"])
    assert o5.heads == {"hm": 1, "proj": 32} and o5.output_h == 32


def test_lr_schedule_matches_reference(golden):
    from cet_pick_amd.utils.utils import adjust_learning_rate

    class A:
        lr, lr_decay_rate, num_epochs, lr_step = 0.02, 0.1, 140, [90, 120]
    opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=0.02)
    for cosine, ep, lr in golden("lr_sched.npz")["rows"]:
        A.cosine = bool(cosine)
        adjust_learning_rate(A, opt, int(ep))
        assert abs(opt.param_groups[0]["lr"] - lr) < 1e-12


def test_factory_and_checkpoint_layout(tmp_path, capsys):
    from cet_pick_amd.models.model import create_model, save_model, load_model
    heads = {"proj": 256, "pred": 256}
    m = create_model("moco3d_18", heads, 0, last_k=3, local_path=None)      # kwargs the reference passes
    keys = json.load(open(os.path.join(HERE, "golden", "ckpt_keys.json")))["moco3d_encoder"]
    sd = m.state_dict()
    assert list(sd) == list(keys) and all(list(sd[k].shape) == keys[k] for k in keys)
    assert m.proj is m.pred
    path = str(tmp_path / "model_last.pth")
    save_model(path, 7, m)
    ck = torch.load(path)
    assert set(ck) == {"epoch", "state_dict"} and ck["epoch"] == 7
    assert all(v.is_contiguous() for v in ck["state_dict"].values())
    # a DataParallel/DDP-style checkpoint ('module.' prefix), one wrong shape and one stray key
    ck2 = {"epoch": 3, "state_dict": {"module." + k: v for k, v in ck["state_dict"].items()}}
    ck2["state_dict"]["module.fc.bias"] = torch.zeros(7)
    ck2["state_dict"]["module.not_there"] = torch.zeros(1)
    torch.save(ck2, path)
    m2 = create_model("moco3d_18", heads, 0)
    before = m2.fc.bias.detach().clone()
    m2 = load_model(m2, path)
    out = capsys.readouterr().out
    assert "Skip loading parameter fc.bias" in out and "Drop parameter not_there" in out
    assert torch.equal(m2.fc.bias, before)
    assert torch.equal(m2.conv1.weight.detach(), m.conv1.weight.detach())
    assert m2.conv1.weight.stride() == m.conv1.weight.stride()            # kernel layout survives loading
    with pytest.raises(NotImplementedError):
        create_model("res_18", {"hm": 1}, 64)                             # exists in the reference, outside the hot path
