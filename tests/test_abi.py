"""The C-ABI library loads and exports every symbol include/cetpick_hip.h declares (CPU only:
no compute calls)."""
import os
import re

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    txt = open(os.path.join(REPO, "include", "cetpick_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(mi_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as ge
    ge.build()
    from cet_pick_amd import _lib
    L = _lib.lib()
    syms = _header_symbols()
    assert len(syms) >= 10
    for s in syms:
        assert hasattr(L, s), "missing export " + s
    assert set(syms) == set(_lib.SIGNATURES), (set(syms) ^ set(_lib.SIGNATURES))
    assert L.mi_abi_version() >= 1
    assert L.mi_build_arch() == b"gfx950"


def test_product_path_has_no_cpu_fallback():
    import numpy as np
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from cet_pick_amd import _lib
    from cet_pick_amd.utils import image as Im
    with pytest.raises(_lib.HipExtensionError):
        Im.get_potential_coords_pyramid(np.zeros((30, 80, 80), np.float32), sigmas=[2, 4])


def test_product_does_not_import_oracle():
    pkg = os.path.join(REPO, "cet_pick_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(root, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
