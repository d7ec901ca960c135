import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_crops_match_reference_semantics():
    from oracle import infer_ref as O
    from cet_pick_amd.datasets import subvols as S
    rng = np.random.default_rng(0)
    vol = rng.standard_normal((40, 90, 100)).astype(np.float32)
    coords = np.stack([rng.integers(20, 80, 50), rng.integers(20, 70, 50), rng.integers(2, 38, 50)], 1).astype(np.int32)
    v = torch.as_tensor(vol).cuda()
    got = S.extract_subvols(v, coords, (3, 36, 36)).cpu().numpy()
    raw = S.extract_subvols_3d(v, coords, (3, 36, 36)).cpu().numpy()
    for i, c in enumerate(coords):
        np.testing.assert_array_equal(raw[i], O.extract_subvols_3d(vol, c, (3, 36, 36)))
        np.testing.assert_allclose(got[i], O.extract_subvols(vol, c, (3, 36, 36)), rtol=0, atol=2e-6)
    zn = S.crop_znorm(v, coords, (32, 32, 32)[:1] + (32, 32)).cpu().numpy() if False else None
    c3 = np.stack([rng.integers(20, 80, 9), rng.integers(20, 70, 9), rng.integers(17, 23, 9)], 1).astype(np.int32)
    a = S.crop_znorm(v, c3, (32, 32, 32)).cpu().numpy()
    b = S.crop_znorm(v, c3, (32, 32, 32), flip_x=True).cpu().numpy()
    for i, (x, y, z) in enumerate(c3):
        ref = vol[z - 16:z + 16, y - 16:y + 16, x - 16:x + 16]
        ref = (ref - ref.mean()) / ref.std(ddof=1)
        np.testing.assert_allclose(a[i, 0], ref, rtol=0, atol=2e-5)
        np.testing.assert_allclose(b[i, 0], ref[:, :, ::-1], rtol=0, atol=2e-5)
    assert S.extract_subvols(v, np.zeros((0, 3), np.int32), (3, 36, 36)).shape == (0, 1, 36, 36)


def test_crops_equal_reference_golden(golden):
    """mi_crop_normalize against the reference's own crop methods (tests/golden/crops.npz): raw crops bit-exact, the
    min-max'ed projections to the last float ulp, the dataset mean / std, and every `cutup` window the fixture samples."""
    import torch
    from cet_pick_amd.datasets import subvols as S
    from cet_pick_amd.synthetic import make_tomo
    g = golden("crops.npz")
    vol, _ = make_tomo((20, 72, 80), seed=41, margin_xy=20, margin_z=4)
    v = torch.as_tensor(vol).cuda()
    for size in ((3, 24, 24), (3, 16, 20), (5, 8, 12)):
        tag = "%d_%d_%d" % size
        sub = S.extract_subvols(v, g["coords"], size)
        np.testing.assert_allclose(sub.cpu().numpy(), g["sub2d_" + tag], rtol=0, atol=2e-7)
        if "sub3d_" + tag in g:
            np.testing.assert_array_equal(S.extract_subvols_3d(v, g["coords"], size).cpu().numpy(), g["sub3d_" + tag])
        mean, std = S.subvol_mean_std(sub)
        assert abs(mean - float(g["mean_" + tag])) < 1e-6 and abs(std - float(g["std_" + tag])) < 1e-6
    np.testing.assert_allclose(S.extract_3d_tomo(v, g["coords"], 24, 16).cpu().numpy(), g["tomo2d_24_16"], rtol=0, atol=2e-7)
    # cutup windows as crops: centre of window (i, j, k), no margin
    centres, inner = S.cutup_centres(vol.shape, (8, 64, 64), (2, 4, 4))
    nb = tuple(int(x) for x in g["cutup_shape"][:3])
    assert len(centres) == nb[0] * nb[1] * nb[2] and inner == (8, 64, 64)
    idx = [int((i * nb[1] + j) * nb[2] + k) for i, j, k in g["cutup_samples_idx"]]
    # (an 8-slab window has an even z extent: the raw mode's window [c - s/2, c - s/2 + s) is exactly the cutup block)
    from cet_pick_amd.datasets.subvols import _crop, RAW
    got = _crop(v, centres[idx], (8, 64, 64), RAW).cpu().numpy()
    np.testing.assert_array_equal(got, g["cutup_samples"])


def test_chain_and_u8_normalize_vs_oracle():
    """datasets/tomo_pre.py:57-60 (Crop -> ZNormalization -> RescaleIntensity(-3,3) -> ZNormalization) and
    simsiam_test_hm_3d.py:45-51 (8-bit round trip + Normalize) against the oracle's restatement (torchio / torchvision are
    absent: parity unpinned for these two)."""
    import torch
    from oracle import infer_ref as O
    from cet_pick_amd.datasets import subvols as S
    from cet_pick_amd.synthetic import make_tomo
    vol, _ = make_tomo((20, 72, 80), seed=41, margin_xy=20, margin_z=4)
    v = torch.as_tensor(vol).cuda()
    size, stride, margin = (8, 64, 64), (2, 4, 4), (1, 8, 8)
    centres, inner = S.cutup_centres(vol.shape, size, stride, margin)
    assert inner == (6, 48, 48)
    got = S.crop_znorm_rescale_znorm(v, centres, inner).cpu().numpy()
    blks = O.cutup(vol, size, stride)
    nb = blks.shape[:3]
    for n in (0, 7, len(centres) // 2, len(centres) - 1):
        i, j, k = np.unravel_index(n, nb)
        np.testing.assert_allclose(got[n, 0], O.znorm_rescale_znorm(blks[i, j, k], margin), rtol=0, atol=5e-6)
    sub = S.extract_subvols(v, centres[:16], (3, 24, 24))
    mean, std = S.subvol_mean_std(sub)
    y = S.to_uint8_normalize(sub, mean, std).cpu().numpy()
    np.testing.assert_allclose(y, O.u8_roundtrip_normalize(sub.cpu().numpy(), mean, std), rtol=0, atol=1e-6)


def test_fast_crop_kernel_equals_the_generic_one(monkeypatch):
    """crop_fast_kernel (one read of the crop, 1024 threads, lanes along x) against the generic kernel (MI_CROP_GENERIC=1) on the
    shapes it takes - 32^3 (MoCo), 6 x 48 x 48 and 6 x 64 x 64 (the 3-D chain's windows), a ragged 5 x 7 x 19 - raw and
    z-normalised, mirrored, with centres at the volume's edges (clamped reads)."""
    import numpy as np
    import torch
    from cet_pick_amd.datasets import subvols as S
    rng = np.random.default_rng(3)
    vol = torch.as_tensor(rng.standard_normal((40, 90, 100)).astype(np.float32)).cuda()
    for size in ((32, 32, 32), (6, 48, 48), (6, 64, 64), (5, 7, 19)):
        cen = np.stack([rng.integers(0, 100, 24), rng.integers(0, 90, 24), rng.integers(0, 40, 24)], 1).astype(np.int32)
        for mode in (S.RAW, S.ZNORM):
            for flip in (False, True):
                a = S._crop(vol, cen, size, mode, flip)
                monkeypatch.setenv("MI_CROP_GENERIC", "1")
                b = S._crop(vol, cen, size, mode, flip)
                monkeypatch.delenv("MI_CROP_GENERIC")
                if mode == S.RAW:
                    assert torch.equal(a, b)
                else:
                    np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=0, atol=3e-6)
