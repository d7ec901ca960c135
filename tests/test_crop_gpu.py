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
