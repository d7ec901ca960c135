"""CPU: the detector-network oracle (oracle/unet_ref.py) against outputs of the reference's own TomoConvUNet, and
state_dict compatibility of the MI355X module with the reference's keys / shapes."""
import json
import os

import numpy as np
import torch

from oracle import unet_ref as O
from cet_pick_amd.synthetic import seeded_state_dict

G = os.path.join(os.path.dirname(__file__), "golden")
HEADS = {"hm": 1, "proj": 32}


def _net():
    from cet_pick_amd.models.networks.unet_small import TomoConvUNet
    return TomoConvUNet(4, HEADS, 32, 3)


def test_state_dict_keys_match_reference():
    keys = json.load(open(os.path.join(G, "ckpt_keys.json")))["unet_4"]
    assert {k: list(v.shape) for k, v in _net().state_dict().items()} == keys


def test_factory_builds_unet():
    from cet_pick_amd.models.model import create_model
    m = create_model("unet_5", HEADS, 32)
    assert m.n_blocks == 5 and len(m.unet.down_convs) == 5 and len(m.unet.up_convs) == 4


def test_oracle_matches_reference_outputs():
    g = np.load(os.path.join(G, "unet4.npz"))
    sd = seeded_state_dict(_net(), seed=321)
    for tag in ("a", "odd", "b2"):
        out = O.tomo_conv_unet_forward(sd, torch.from_numpy(g[f"x_{tag}"]), 4, HEADS)
        np.testing.assert_allclose(out["hm"].numpy(), g[f"hm_{tag}"], rtol=0, atol=1e-5)
        np.testing.assert_allclose(out["proj"].numpy()[:, :, :, ::3, ::3], g[f"proj_{tag}"], rtol=0, atol=1e-6)


def test_cpu_tensor_fails_loudly():
    import pytest
    from cet_pick_amd._lib import HipExtensionError
    with pytest.raises(HipExtensionError):
        _net()(torch.zeros(1, 2, 16, 16))
