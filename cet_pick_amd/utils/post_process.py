"""Mirror of the reference's `cet_pick/utils/post_process.py` host-side post-processing (≤ K detections, numpy):
`tomo_post_process` :11-25 (group detections by integer z), `tomo_group_postprocess` :31-50 and
`tomo_fiber_postprocess` :52-113 (distance-graph components; quadratic fits along y).  The connected components
come from scipy.sparse.csgraph (the reference's `sknetwork` is an un-vendored third-party package).
"""
import numpy as np
from scipy import sparse
from scipy.sparse.csgraph import connected_components


def tomo_post_process(dets, z_dim_tot=128):
    """dets (batch, K, 5) numpy [x, y, z, score, score] -> [{z: [[x, y, z, s, s], ...]}] (post_process.py:11-25;
    like the reference only the LAST batch element's dict is returned inside the list)."""
    ret = []
    top_preds = {}
    for i in range(dets.shape[0]):
        top_preds = {}
        z_dim = dets[i, :, 2]
        for j in range(z_dim_tot):
            inds = z_dim == j
            if inds.any():
                top_preds[j] = dets[i, inds, :].astype(np.float32).tolist()
    ret.append(top_preds)
    return ret


def _components(points, distance_cutoff):
    pts = np.asarray(points, dtype=np.float64)[:, :3]
    d2 = ((pts[:, None, :] - pts[None, :, :]) ** 2).sum(-1)
    adj = sparse.csr_matrix(np.sqrt(d2) <= distance_cutoff)
    return connected_components(adj, directed=False)[1]


def tomo_group_postprocess(dets_all, distance_cutoff=15, min_per_group=5):
    """post_process.py:31-50: keep detections whose distance-graph component has more than min_per_group members."""
    dets = np.asarray(dets_all)
    if dets.size == 0:
        return []
    labels = _components(dets, distance_cutoff)
    out = []
    for lb in np.unique(labels):
        members = dets[labels == lb]
        if members.shape[0] > min_per_group:
            out.extend(list(members))
    return out


def k_x(y, a, b, c):
    """post_process.py:27-29 (the exponent 2/3 is the reference's)."""
    return np.max((2 * a) / ((1 + (2 * a * y + b) ** 2)) ** (2 / 3))


def tomo_fiber_postprocess(dets, distance_cutoff=15, res_cutoff=30, curvature_cutoff=0.03, scale=2):
    """post_process.py:52-113: components of > 6 picks -> quadratic fits x(y), z(y); resample along y every
    `scale` pixels when the fit residual and curvature are small.  Input rows are [x, y, z]; output [y', z, x]
    order follows the reference (columns 0 and 1 are swapped before fitting)."""
    dets = np.asarray(dets, dtype=np.float64)
    out = []
    if dets.size == 0:
        return out
    labels = _components(dets, distance_cutoff)
    for lb in np.unique(labels):
        line = dets[labels == lb].copy()
        if line.shape[0] <= 6:
            continue
        line[:, [1, 0]] = line[:, [0, 1]]
        lo, hi = np.min(line[:, 1]), np.max(line[:, 1])
        span = hi - lo
        y_range = np.linspace(lo - 1, hi + 1, int(span // 2))
        y_out = np.linspace(lo - 1, hi + 1, int(span // scale))
        if y_range.shape[0] == 0:
            continue
        n_fit = line.shape[0]
        fit_x = np.polyfit(line[:, 1], line[:, 0], 2, full=True)
        fit_z = np.polyfit(line[:, 1], line[:, 2], 2, full=True)
        res_x = fit_x[1][0] / n_fit if fit_x[1].shape[0] > 0 else 10000
        res_z = fit_z[1][0] / n_fit if fit_z[1].shape[0] > 0 else 10000
        kx, kz = k_x(y_range, *fit_x[0]), k_x(y_range, *fit_z[0])
        tot = res_x + res_z
        ok = (tot < res_cutoff and abs(kx) < curvature_cutoff and abs(kz) < curvature_cutoff) or \
             (res_cutoff <= tot < res_cutoff * 3 and abs(kx) < curvature_cutoff / 10 and abs(kz) < curvature_cutoff / 10)
        if ok:
            x_fit, z_fit = np.polyval(fit_x[0], y_out), np.polyval(fit_z[0], y_out)
            out.extend([int(y_out[j]), int(z_fit[j]), int(x_fit[j])] for j in range(x_fit.shape[0]))
    return out
