"""The reference's entry points for the train path, end to end on the GPU: flags -> create_model ->
MoCo -> train_factory -> trainer.train(epoch, loader) -> save_model / load_model."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_moco_main_two_epochs_and_resume(tmp_path, monkeypatch):
    from cet_pick_amd.opts import opts
    from cet_pick_amd import moco_main
    from cet_pick_amd.models.model import create_model, load_model
    from cet_pick_amd.models.moco import MoCo
    monkeypatch.chdir(tmp_path)
    args = ["moco", "--arch", "moco3d_18", "--batch_size", "16", "--num_epochs", "2", "--lr", "0.01", "--lr_step", "1",
            "--exp_id", "t", "--debug", "0", "--print_iter", "4", "--val_intervals", "2"]
    moco_main.main(opts().parse(args))
    save_dir = os.path.join(str(tmp_path), "exp", "moco", "t")
    lines = open(os.path.join(save_dir, "log.txt")).read().strip().split("\n")
    assert len(lines) == 2 and lines[0].startswith("epoch: 1 |loss ") and "infoNCE" in lines[0] and "time" in lines[0]
    losses = [float(l.split("|")[1].split()[1]) for l in lines]
    assert all(np.isfinite(losses)) and losses[1] < losses[0] + 0.5
    assert os.path.exists(os.path.join(save_dir, "model_last_contrastive.pth"))      # epoch 1
    ck = torch.load(os.path.join(save_dir, "model_last.pth"))                          # epoch 2 (val interval)
    assert set(ck) == {"epoch", "state_dict", "optimizer"} and ck["epoch"] == 2
    assert len(ck["state_dict"]) == 122 and "queue" in ck["state_dict"] and "encoder_k.fc.weight" in ck["state_dict"]
    assert os.path.exists(os.path.join(save_dir, "model_1.pth"))
    heads = {"proj": 256, "pred": 256}
    m = MoCo(create_model("moco3d_18", heads, 0), create_model("moco3d_18", heads, 0), dim=128)
    m = load_model(m, os.path.join(save_dir, "model_last.pth"))
    assert torch.equal(m.queue, ck["state_dict"]["queue"]) and int(m.queue_ptr) == int(ck["state_dict"]["queue_ptr"])
    # the key encoder trails the query encoder (EMA), neither is the initial state
    assert not torch.equal(m.encoder_q.fc.weight, m.encoder_k.fc.weight)
    # resume continues from the stored epoch with the decayed lr
    moco_main.main(opts().parse(args[:5] + ["--num_epochs", "3", "--lr", "0.01", "--lr_step", "1", "--exp_id", "t",
                                            "--debug", "0", "--resume"]))
    assert len(open(os.path.join(save_dir, "log.txt")).read().strip().split("\n")) == 3
