"""Synthetic tomograms and seeded weights shared by tests, bench and the golden-vector generator.

SURVEY.md §8(d) defines the synthetic tomogram: fp32 N(0,1) noise plus Gaussian blobs whose
centres keep a minimum pairwise distance so greedy-NMS results are tie-free.  Stored (Z, H, W),
i.e. the in-memory order the reference holds after ``load_rec`` (utils/loader.py:27-88).
"""
import numpy as np


def make_tomo(shape, seed=317, blob_spacing=16, margin_xy=40, margin_z=12, dtype=np.float32):
    """Return (volume (Z,H,W), centres (n,3) as x,y,z)."""
    z, h, w = shape
    rng = np.random.default_rng(seed)
    vol = rng.standard_normal(shape, dtype=np.float32)
    n_target = max(1, int(z * h * w / 64 ** 3))
    mz = min(margin_z, max(0, z // 2 - 1))
    my = min(margin_xy, max(0, h // 2 - 1))
    mx = min(margin_xy, max(0, w // 2 - 1))
    centres = []
    # jittered grid keeps the pairwise distance >= blob_spacing by construction
    cell = max(blob_spacing * 2, 32)
    gz = np.arange(mz, max(mz + 1, z - mz), cell)
    gy = np.arange(my, max(my + 1, h - my), cell)
    gx = np.arange(mx, max(mx + 1, w - mx), cell)
    grid = np.stack(np.meshgrid(gz, gy, gx, indexing="ij"), -1).reshape(-1, 3)
    rng.shuffle(grid)
    jit = cell - blob_spacing
    for cz, cy, cx in grid[:n_target]:
        dz, dy, dx = rng.integers(0, max(1, jit), size=3)
        pz = int(min(cz + dz, z - mz - 1)) if z - mz - 1 >= mz else int(cz)
        py = int(min(cy + dy, h - my - 1)) if h - my - 1 >= my else int(cy)
        px = int(min(cx + dx, w - mx - 1)) if w - mx - 1 >= mx else int(cx)
        sigma = float(rng.choice([2.0, 3.0, 4.0]))
        amp = float(rng.uniform(2.0, 4.0))
        r = int(4 * sigma)
        z0, z1 = max(0, pz - r), min(z, pz + r + 1)
        y0, y1 = max(0, py - r), min(h, py + r + 1)
        x0, x1 = max(0, px - r), min(w, px + r + 1)
        zz, yy, xx = np.ogrid[z0:z1, y0:y1, x0:x1]
        g = np.exp(-((zz - pz) ** 2 + (yy - py) ** 2 + (xx - px) ** 2) / (2 * sigma * sigma))
        vol[z0:z1, y0:y1, x0:x1] += (amp * g).astype(np.float32)
        centres.append((px, py, pz))
    return vol.astype(dtype, copy=False), np.asarray(centres, dtype=np.int32).reshape(-1, 3)


def make_logits(shape, seed=317, n_peaks=None):
    """Synthetic detector logit volume (D,H,W): background ~N(-4,1) with sparse positive peaks."""
    rng = np.random.default_rng(seed)
    d, h, w = shape
    vol = (rng.standard_normal(shape, dtype=np.float32) - 4.0).astype(np.float32)
    if n_peaks is None:
        n_peaks = max(4, d * h * w // 4096)
    flat = rng.choice(d * h * w, size=n_peaks, replace=False)
    vol.reshape(-1)[flat] = rng.uniform(0.0, 6.0, size=n_peaks).astype(np.float32)
    return vol


def seeded_state_dict(model, seed=317, gain=1.0):
    """Deterministic O(1)-activation weights for any nn.Module (fan-in scaled normal).

    The reference's default init draws feature_3d/fc/proj weights with std 1e-3
    (moco_encoder_3d.py:137-154), which leaves BatchNorm dividing by sqrt(eps); parity fixtures use
    this well-conditioned init instead, regenerated identically in the tests from the seed alone.
    """
    import torch
    g = torch.Generator().manual_seed(seed)
    out = {}
    for k, v in model.state_dict().items():
        if k.endswith("num_batches_tracked"):
            out[k] = torch.zeros_like(v)
        elif k.endswith("running_mean"):
            out[k] = torch.randn(v.shape, generator=g) * 0.1
        elif k.endswith("running_var"):
            out[k] = torch.rand(v.shape, generator=g) * 0.5 + 0.75
        elif v.dim() == 1:
            if k.endswith("bias"):
                out[k] = torch.randn(v.shape, generator=g) * 0.1
            else:
                out[k] = torch.rand(v.shape, generator=g) * 0.5 + 0.75
        else:
            fan_in = v[0].numel()
            out[k] = torch.randn(v.shape, generator=g) * (gain * (2.0 / fan_in) ** 0.5)
    return out


def losses_inputs(seed=31, n=512, dim=32, shape=(2, 1, 4, 16, 16)):
    """Seeded inputs of the detector-loss fixtures (tests/golden/losses.npz): heat-map / label volumes with positive (1),
    soft, labelled-negative (0) and unlabeled (-1) voxels, two views of L2-normalised features, per-voxel labels and
    predictions with values beyond the 0.99 / 0.01 pseudo-label thresholds."""
    import torch
    g = torch.Generator().manual_seed(seed)
    pred = torch.rand(shape, generator=g).clamp(1e-4, 1 - 1e-4)
    gt = torch.full(shape, -1.0)
    r = torch.rand(shape, generator=g)
    gt[r < 0.30] = 0.0
    soft = (r >= 0.30) & (r < 0.40)
    gt[soft] = (torch.rand(shape, generator=g)[soft] * 0.9)
    gt[r > 0.97] = 1.0
    f = torch.nn.functional.normalize(torch.randn(n, dim, generator=g), dim=1)
    f_cr = torch.nn.functional.normalize(f + 0.3 * torch.randn(n, dim, generator=g), dim=1)
    lab = torch.full((n,), -1.0)
    rr = torch.rand(n, generator=g)
    lab[rr < 0.25] = 0.0
    lab[(rr >= 0.25) & (rr < 0.35)] = 0.5
    lab[rr > 0.95] = 1.0
    o1 = torch.rand(n, generator=g)
    o1[torch.rand(n, generator=g) < 0.15] = 0.995
    o1[torch.rand(n, generator=g) < 0.15] = 0.004
    o2 = (o1 + 0.05 * torch.randn(n, generator=g)).clamp(1e-4, 1 - 1e-4)
    return pred, gt, f, f_cr, lab, o1, o2


def confident_pred(gt, seed=37):
    """A heat-map that agrees with the labels (high on positives, low elsewhere): drives the PU risk into its
    `neg_risk_total < -beta` branch (loss.py:305-306)."""
    import torch
    g = torch.Generator().manual_seed(seed)
    r = torch.rand(gt.shape, generator=g)
    return torch.where(gt == 1, 0.90 + 0.09 * r, 0.02 + 0.05 * r)


def semi_loss_inputs(flip_prob, shape=(2, 1, 3, 12, 16), dim=32):
    """Seeded inputs of the `TomoCRSemiLoss` fixture (tests/golden/semi_loss.npz): label volume with unlabeled (-1),
    labelled-negative (0), soft (0.6) and positive (1) voxels, heat-map logits and L2-normalised projections of both
    views.  Returns (gt, hm_logits, hm_logits_cr, proj, proj_cr)."""
    import torch
    g = torch.Generator().manual_seed(int(flip_prob * 10))
    gt = torch.full(shape, -1.0)
    r = torch.rand(shape, generator=g)
    gt[r < 0.3] = 0.0
    gt[(r >= 0.3) & (r < 0.4)] = 0.6
    gt[r > 0.96] = 1.0
    hm, hm_cr = torch.randn(shape, generator=g), torch.randn(shape, generator=g)
    pshape = (shape[0], dim) + tuple(shape[2:])
    pj = torch.nn.functional.normalize(torch.randn(pshape, generator=g), dim=1)
    pj_cr = torch.nn.functional.normalize(torch.randn(pshape, generator=g), dim=1)
    return gt, hm, hm_cr, pj, pj_cr


def moco_small_inputs(batch=8, dim=128, K=64):
    """Seeded inputs of the symmetric-MoCo fixture (tests/golden/moco_small.npz): two views and the initial queue."""
    import torch
    g = torch.Generator().manual_seed(12)
    im1 = torch.randn(batch, 1, 32, 32, 32, generator=g)
    im2 = torch.randn(batch, 1, 32, 32, 32, generator=g)
    queue0 = torch.nn.functional.normalize(torch.randn(dim, K, generator=g), dim=0)
    return im1, im2, queue0
