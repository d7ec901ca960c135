"""CPU: host-side post-processing mirrors (utils/post_process.py of the reference)."""
import numpy as np

from cet_pick_amd.utils.post_process import tomo_fiber_postprocess, tomo_group_postprocess, tomo_post_process


def test_tomo_post_process_groups_by_integer_z():
    dets = np.array([[[1.5, 2.5, 3.0, 0.9, 0.9], [4.0, 5.0, 3.0, 0.8, 0.8], [7.0, 8.0, 5.0, 0.7, 0.7], [0, 0, 4.5, 0.1, 0.1]]],
                    dtype=np.float32)
    out = tomo_post_process(dets, z_dim_tot=6)
    assert len(out) == 1 and sorted(out[0]) == [3, 5]            # 4.5 is not an integer z: dropped like the reference
    assert len(out[0][3]) == 2 and out[0][5][0][:3] == [7.0, 8.0, 5.0]
    assert tomo_post_process(dets, z_dim_tot=4)[0].keys() == {3}


def test_group_postprocess_keeps_large_components():
    rng = np.random.default_rng(0)
    a = rng.normal(0, 2, (8, 3)) + [50, 50, 20]
    b = rng.normal(0, 2, (3, 3)) + [150, 150, 20]
    pts = np.concatenate([np.c_[a, np.ones(8)], np.c_[b, np.ones(3)]])
    out = np.array(tomo_group_postprocess(pts, distance_cutoff=15, min_per_group=5))
    assert out.shape == (8, 4) and np.all(out[:, 0] < 100)
    assert tomo_group_postprocess([], 15) == []


def test_fiber_postprocess_resamples_a_straight_fiber():
    # rows are [x, y, z]; the reference swaps columns 0/1 and fits y(x), z(x): output rows are [x', z, y']
    x = np.arange(20, 100, 6, dtype=np.float64)
    line = np.stack([x, 30 + 0.1 * x, np.full_like(x, 12.0)], 1)
    noise = np.array([[200.0, 200.0, 5.0]])
    out = np.array(tomo_fiber_postprocess(np.concatenate([line, noise]), distance_cutoff=15, res_cutoff=30,
                                          curvature_cutoff=0.03, scale=2))
    assert out.shape == (int((x.max() - x.min()) // 2), 3)
    assert np.all(np.abs(out[:, 1] - 12) <= 1)                            # z column
    assert np.all(np.abs(out[:, 2] - (30 + 0.1 * out[:, 0])) <= 2)        # y(x) on the fitted line
