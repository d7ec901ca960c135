"""The inference oracle pinned against vectors produced by the reference itself
(tests/golden/gen_golden.py).  CPU only."""
import numpy as np

from oracle import infer_ref as O
from cet_pick_amd.synthetic import make_tomo


def test_sigmoid_and_nms_windows(golden):
    g = golden("decode_small.npz")
    hm = O.sigmoid_clamp(g["logits"])
    np.testing.assert_allclose(hm, g["sigmoid"], rtol=0, atol=2e-7)
    hm = g["sigmoid"]  # continue from the reference's own bits so equality masks are exact
    for k in (3, 5):
        np.testing.assert_array_equal(O.nms_window(hm, (3, k, k)), g[f"nms_3kk_{k}"])
        np.testing.assert_array_equal(O.nms_window(hm, (1, k, k)), g[f"nms_xy_{k}"])
        np.testing.assert_array_equal(O.nms_window(hm, (k, 1, 1)), g[f"nms_z_{k}"])
        np.testing.assert_array_equal(O.nms_window(hm, (k, k, k)), g[f"nms_kkk_{k}"])


def _same_dets(a, b):
    # compare as sets of rows above the clamp floor (ties at the floor have no defined order)
    fa = a[a[:, 3] > 1.5e-4]
    fb = b[b[:, 3] > 1.5e-4]
    assert len(fa) == len(fb)
    np.testing.assert_array_equal(fa, fb)


def test_tomo_decode(golden):
    g = golden("decode_small.npz")
    hm = g["sigmoid"]
    for k in (3, 5):
        _same_dets(O.tomo_decode(hm, kernel=k, K=50), g[f"decode_{k}"])
        _same_dets(O.tomo_decode(hm, kernel=k, K=50, if_fiber=True), g[f"decode_fiber_{k}"])
    s, z, y, x, inds = O.topk(O.nms_window(hm, (3, 3, 3)), 40)
    m = g["topk_scores"] > 1.5e-4
    np.testing.assert_array_equal(s[m], g["topk_scores"][m])
    np.testing.assert_array_equal(inds[m], g["topk_inds"][m])
    np.testing.assert_array_equal(z[m], g["topk_z"][m])
    np.testing.assert_array_equal(y[m], g["topk_y"][m])
    np.testing.assert_array_equal(x[m], g["topk_x"][m])
    # image.py variant: x = t % h (utils/image.py:111), pinned on H != W
    d, h, w = hm.shape
    z2, y2, x2 = O.convert_1d_to_3d(inds, d, h, w, image_variant=True)
    np.testing.assert_array_equal(x2[m], g["img_topk_x"][m])


def test_greedy_nms(golden):
    g = golden("greedy_small.npz")
    vol = g["vol"]
    for d, thr in ((6, 0.5), (14, 1.0), (4, -np.inf)):
        s, c = O.non_maximum_suppression_3d(vol, d, threshold=thr)
        np.testing.assert_array_equal(s, g[f"scores_d{d}"])
        np.testing.assert_array_equal(c, g[f"coords_d{d}"])
    s, c = O.non_maximum_suppression_3d(vol.astype(np.float64), 6, scale=1.5, threshold=0.2)
    np.testing.assert_array_equal(s, g["scores_d6_s15"])
    np.testing.assert_array_equal(c, g["coords_d6_s15"])


def test_gaussian_matches_scipy_fixture(golden):
    g = golden("dog_small.npz")
    shape = tuple(int(v) for v in g["shape"])
    vol, centres = make_tomo(shape, seed=317)
    np.testing.assert_array_equal(centres, g["centres"])
    rec = vol.astype(np.float64)
    for sig in (2, 3, 5):
        f = O.gaussian_filter(rec, sig)
        np.testing.assert_allclose(f[18, ::3], g[f"gauss{sig}_z18"], rtol=0, atol=1e-12)
        np.testing.assert_allclose(f[0, ::3], g[f"gauss{sig}_z0"], rtol=0, atol=1e-12)


def test_potential_coords_pyramid(golden):
    g = golden("dog_small.npz")
    shape = tuple(int(v) for v in g["shape"])
    vol, _ = make_tomo(shape, seed=317)
    rec = vol.astype(np.float64)
    for sigmas in ((2, 4), (3, 5), (2, 4, 6)):
        tag = "_".join(map(str, sigmas))
        s, c = O.get_potential_coords_pyramid(rec, sigmas=sigmas)
        assert len(s) == len(g[f"scores_{tag}"]) and len(s) > 0
        np.testing.assert_array_equal(c, g[f"coords_{tag}"])
        np.testing.assert_allclose(s, g[f"scores_{tag}"], rtol=1e-6)
    s, c = O.get_potential_coords_pyramid(vol, sigmas=(3, 5), dtype=np.float32)
    np.testing.assert_array_equal(c, g["coords_3_5_f32in"])
