"""CPU: the loader oracle (oracle/preproc_ref.py) against vectors produced by the reference's own
utils/loader.py, and the dependency-free MRC reader / writer against a file the reference's utils/mrc.py wrote."""
import os

import numpy as np
import pytest

from oracle import preproc_ref as O
from cet_pick_amd.utils import mrc

G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(G, "loader_small.npz"))


@pytest.mark.parametrize("order", ["xyz", "xzy", "yxz", "zxy"])
@pytest.mark.parametrize("compress", [False, True])
def test_load_rec_matches_reference(gold, order, compress):
    got = O.load_rec(gold["vol"], order, compress)
    ref = gold[f"load_{order}_{int(compress)}"]
    assert got.shape == ref.shape and got.dtype == np.float64
    np.testing.assert_allclose(got, ref, rtol=0, atol=1e-12)


@pytest.mark.parametrize("order", ["xyz", "xzy", "yxz", "zxy"])
def test_load_rec_tilt_matches_reference(gold, order):
    np.testing.assert_allclose(O.load_rec(gold["vol"], order, True, is_tilt=True), gold[f"load_tilt_{order}"],
                               rtol=0, atol=1e-5)     # per-slice statistics of float32 slices


def test_load_rec_int16(gold):
    np.testing.assert_allclose(O.load_rec(gold["vol_i16"], "xzy", True), gold["load_i16_xzy_1"], rtol=0, atol=1e-12)


def test_zxy_compress_odd_is_an_error():
    with pytest.raises(IndexError):
        O.load_rec(np.zeros((5, 4, 4), np.float32), "zxy", True)


def test_preprocess_and_quantize(gold):
    z = gold["load_xzy_0"]
    np.testing.assert_array_equal(O.quantize(z), gold["quant"])
    np.testing.assert_array_equal(O.quantize(z, mi=-3, ma=3), gold["quant_33"])
    np.testing.assert_array_equal(O.quantize(z, mi=None, ma=None), gold["quant_auto"])
    np.testing.assert_allclose(O.preprocess(z, 0), gold["pre_0"], rtol=0, atol=1e-15)
    np.testing.assert_allclose(O.preprocess(z, 1.0), gold["pre_dn"], rtol=0, atol=1e-15)


def test_reader_on_reference_written_file(gold):
    arr, hdr = mrc.parse_mrc(os.path.join(G, "ref_written.mrc"))
    np.testing.assert_array_equal(arr, gold["ref_written_data"])
    f = [hdr[k] for k in ("nx", "ny", "nz", "mode", "mapc", "mapr", "maps", "ispg", "next")]
    np.testing.assert_array_equal(np.array(f), gold["ref_written_fields"])
    np.testing.assert_array_equal(mrc.open_data(os.path.join(G, "ref_written.mrc"), mmap=True), arr)


def test_writer_byte_identical_to_reference(tmp_path, gold):
    # same array -> same 1024-byte header + payload as the file the reference wrote
    p = tmp_path / "w.mrc"
    mrc.write(str(p), gold["ref_written_data"])
    assert p.read_bytes() == open(os.path.join(G, "ref_written.mrc"), "rb").read()


@pytest.mark.parametrize("dtype", [np.int8, np.int16, np.uint16, np.float32])
def test_write_parse_round_trip(tmp_path, dtype):
    a = (np.random.default_rng(1).standard_normal((3, 5, 4)) * 50).astype(dtype)
    p = str(tmp_path / "a.mrc")
    mrc.write(p, a)
    b, h = mrc.parse_mrc(p)
    assert b.dtype == np.dtype(dtype) and np.array_equal(a, b)
    lazy, _ = mrc.parse_mrc(p, lazy=True)
    assert np.array_equal(lazy[2].get(), a[2])
    h.update_apix(2.5)
    assert abs(h.get_apix() - 2.5) < 1e-6


def test_truncated_file_is_read_permissively(tmp_path):
    a = np.arange(4 * 3 * 2, dtype=np.float32).reshape(4, 3, 2)
    p = tmp_path / "t.mrc"
    mrc.write(str(p), a)
    raw = p.read_bytes()
    p.write_bytes(raw[:-3 * 2 * 4 - 5])          # lose the last section and a bit
    b, _ = mrc.parse_mrc(str(p))
    assert np.array_equal(b, a[:2])
