"""GPU parity of the device loader (cet_pick_amd/utils/loader.py -> csrc/preproc.hip) with the reference's
loader.py (golden vectors) and the float64 oracle on larger seeded volumes."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(G, "loader_small.npz"))


def _np(t):
    return t.cpu().numpy().astype(np.float64)


@pytest.mark.parametrize("order", ["xyz", "xzy", "yxz", "zxy"])
@pytest.mark.parametrize("compress", [False, True])
def test_load_rec_golden(gold, order, compress):
    from cet_pick_amd.utils import loader
    got = loader.load_rec(gold["vol"], order, compress)
    ref = gold[f"load_{order}_{int(compress)}"]
    assert tuple(got.shape) == ref.shape and got.dtype == torch.float32
    np.testing.assert_allclose(_np(got), ref, rtol=0, atol=5e-7)          # fp32 storage of O(1) values


def test_load_rec_from_file_and_int16(gold, tmp_path):
    from cet_pick_amd.utils import loader, mrc
    p = str(tmp_path / "v.mrc")
    mrc.write(p, gold["vol_i16"])
    np.testing.assert_allclose(_np(loader.load_rec(p, "xzy", True)), gold["load_i16_xzy_1"], rtol=0, atol=5e-7)
    np.testing.assert_allclose(_np(loader.load_rec(gold["vol"], "yxz", True, is_tilt=True)), gold["load_tilt_yxz"],
                               rtol=0, atol=2e-5)
    with pytest.raises(IndexError):
        loader.load_rec(np.zeros((5, 4, 4), np.float32), "zxy", True)


def test_preprocess_golden(gold):
    from cet_pick_amd.utils import loader
    z = gold["load_xzy_0"].astype(np.float32)
    got = _np(loader.preprocess(z, 0))
    # levels are k/(qmax-qmin): bit-exact quantisation means every voxel lands on the reference's level
    np.testing.assert_allclose(got, gold["pre_0"], rtol=0, atol=1e-7)
    q = loader.quantize(torch.from_numpy(z).cuda()).cpu().numpy()
    assert np.mean(q != gold["quant"]) < 2e-3       # float32 copy of the float64 input: boundary cases only
    # denoised branch: fp32 separable Gaussian -> a voxel may fall on the neighbouring level
    gd = _np(loader.preprocess(z, 1.0))
    step = 1.0 / 255
    assert np.max(np.abs(gd - gold["pre_dn"])) <= 2 * step and np.mean(np.abs(gd - gold["pre_dn"]) > 1e-6) < 0.02


@pytest.mark.parametrize("shape,order", [((64, 70, 33), "xyz"), ((40, 96, 130), "xzy"), ((31, 50, 65), "yxz"),
                                         ((48, 64, 80), "zxy")])
def test_load_rec_and_preprocess_vs_oracle(shape, order):
    from oracle import preproc_ref as O
    from cet_pick_amd.utils import loader
    rng = np.random.default_rng(sum(shape))
    vol = (rng.standard_normal(shape) * 7 + 100).astype(np.float32)
    for comp in (False, True):
        ref = O.load_rec(vol, order, comp)
        got = loader.load_rec(vol, order, comp)
        np.testing.assert_allclose(_np(got), ref, rtol=0, atol=1e-6)
        pre = _np(loader.preprocess(got, 0))
        pref = O.preprocess(_np(got), 0)           # oracle on exactly the fp32 values the device holds
        bad = np.abs(pre - pref) > 1e-7
        assert bad.mean() < 1e-5, bad.sum()        # ties at a rounding boundary only


def test_tilt_branches_vs_oracle():
    from oracle import preproc_ref as O
    from cet_pick_amd.utils import loader
    rng = np.random.default_rng(9)
    ts = (rng.standard_normal((21, 72, 88)) * 20 + 5).astype(np.float32)
    ts[3] = 4.0                                       # a constant slice: cv2.normalize gives zeros
    got = loader.load_rec(ts, "zxy", False, is_tilt=True)
    ref = O.load_rec(ts, "zxy", False, is_tilt=True)
    ok = np.arange(21) != 3
    np.testing.assert_allclose(_np(got)[ok], ref[ok], rtol=0, atol=2e-5)
    ts2 = ts.copy(); ts2[3] = ts[4]
    pre = _np(loader.preprocess(ts2, 0, is_tilt=True))
    pref = O.preprocess(ts2, 0, is_tilt=True).astype(np.float64)
    assert (np.abs(pre - pref) > 1e-6).mean() < 1e-4
    flat = _np(loader.preprocess(ts, 0, is_tilt=True))[3]
    assert np.all(flat == 0) or np.all(np.isnan(flat))      # z-score of a constant slice is 0/0 upstream
    dn = _np(loader.preprocess(ts2, 1.5, is_tilt=True))
    dref = O.preprocess(ts2, 1.5, is_tilt=True).astype(np.float64)
    assert np.max(np.abs(dn - dref)) <= 2.0 / 255 + 1e-6


def test_cutup_matches_numpy():
    from cet_pick_amd.utils import loader
    a = np.arange(8 * 20 * 24, dtype=np.float32).reshape(8, 20, 24)
    ref = loader.cutup(a, (4, 8, 8), (2, 4, 4))
    got = loader.cutup(torch.from_numpy(a).cuda(), (4, 8, 8), (2, 4, 4))
    assert tuple(got.shape) == ref.shape
    np.testing.assert_array_equal(got.cpu().numpy(), ref)
