import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # a test marked gpu that is collected on a box without a GPU is skipped, never silently passed
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return load


def f32_equivalent(got, cpu32, ref64, factor=2.0, floor=2e-6, what="", more_cpu32=()):
    """fp64-arbitrated equivalence, in place of a widened tolerance: the GPU result may sit as far
    from the float64 evaluation of the same arithmetic as `factor` x the CPU fp32 evaluation does (both as norm-relative
    errors), + `floor` for results that are rounding noise on both sides.  Returns (err_gpu, err_cpu32).
    more_cpu32: further fp32 evaluations of the same quantity (e.g. with inputs perturbed by one fp32 rounding) - the
    arbiter is then the worst of them: where the computation is ill-conditioned (a batch-8 BatchNorm behind exploding
    activations) ONE fp32 evaluation is a heavy-tailed sample of the fp32 error, not its size."""
    g = np.asarray(got, np.float64).ravel()
    c = np.asarray(cpu32, np.float64).ravel()
    r = np.asarray(ref64, np.float64).ravel()
    scale = float(np.linalg.norm(r)) + 1e-30
    e_g, e_c = float(np.linalg.norm(g - r)) / scale, float(np.linalg.norm(c - r)) / scale
    for m in more_cpu32:
        e_c = max(e_c, float(np.linalg.norm(np.asarray(m, np.float64).ravel() - r)) / scale)
    assert e_g <= factor * e_c + floor, "%s: GPU %.3e from float64, CPU fp32 %.3e" % (what, e_g, e_c)
    return e_g, e_c


def gpu_relu_decisions(enc, x):
    """The ReLU decisions of one training-mode forward pass of `enc` on x: {oracle name: bool (N, C, ...)}.  The pass runs
    without autograd and leaves the module's BatchNorm buffers as they were."""
    import torch
    from cet_pick_amd import hipops as H
    saved = {k: v.clone() for k, v in enc.state_dict().items()}
    H.RELU_TAP = {}
    try:
        with torch.no_grad():
            enc(x)
        tap = H.RELU_TAP
    finally:
        H.RELU_TAP = None
    enc.load_state_dict(saved)
    names = {id(m): n for n, m in enc.named_modules()}
    cl = lambda t: (t > 0).permute(0, 4, 1, 2, 3).contiguous().cpu() if t.dim() == 5 else (t > 0).cpu()
    masks = {}
    for k, v in tap.items():
        if isinstance(k, int):
            masks[names[k] + ".mid"], masks[names[k] + ".out"] = cl(v[0]), cl(v[1])
        else:
            masks[k] = cl(v)
    return masks


def assert_relu_flips_on_edge(masks, pre64, pre32=None, max_units=8, rel=1e-4):
    """masks: the GPU's ReLU decisions (gpu_relu_decisions); pre64: the float64 oracle's ReLU inputs (encoder_forward's
    `pre`).  Where the two disagree the unit must be ON THE EDGE: |float64 input| <= rel x the layer's rms - or, where the
    forward pass itself is ill-conditioned (pre32: the CPU fp32 oracle's inputs of the same pass), <= 4 x the largest
    fp32-vs-float64 difference in that layer: a decision is only the GPU's to take where fp32 cannot resolve it.  At most
    `max_units` per layer.  Returns {layer: number of units that differ}."""
    import torch
    flips = {}
    for name, m in masks.items():
        p64 = pre64[name + ".pre"].detach()
        assert p64.shape == m.shape, name
        diff = m != (p64 > 0)
        flips[name] = int(diff.sum())
        if flips[name]:
            # the edge is PER UNIT: the layer-wide rms bound, or - where the forward pass itself is ill-conditioned - four
            # times THIS unit's own fp32-vs-float64 difference (one ill-conditioned unit does not widen the others' bound)
            edge = torch.full_like(p64[diff], rel * float(p64.pow(2).mean().sqrt()))
            if pre32 is not None:
                edge = torch.maximum(edge, 4.0 * (pre32[name + ".pre"].detach().double() - p64).abs()[diff])
            assert flips[name] <= max_units, (name, flips[name])
            over = p64[diff].abs() - edge
            assert float(over.max()) <= 0.0, (name, float(p64[diff].abs().max()), float(edge.min()), int((over > 0).sum()))
    return flips
