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
