"""GPU parity of the inference path (through the C-ABI) against the CPU oracle and the golden
vectors generated from the reference.  Integer/index results must be exact; fp32 scores 1e-5
relative (sigmoid/Gaussian arithmetic differs in the last ulps between libm and the GPU)."""
import numpy as np
import pytest
import torch

from cet_pick_amd.synthetic import make_tomo, make_logits

pytestmark = pytest.mark.gpu


def dev(x):
    return torch.as_tensor(np.ascontiguousarray(x)).cuda()


def test_sigmoid_inplace_semantics(golden):
    from cet_pick_amd.models.utils import _sigmoid
    g = golden("decode_small.npz")
    x = dev(g["logits"])[None, None].clone()
    y = _sigmoid(x)
    np.testing.assert_allclose(y[0, 0].cpu().numpy(), g["sigmoid"], rtol=2e-6, atol=1e-7)
    # input is overwritten with the un-clamped sigmoid (reference mutates x)
    ref = 1.0 / (1.0 + np.exp(-g["logits"].astype(np.float64)))
    np.testing.assert_allclose(x[0, 0].cpu().numpy(), ref, rtol=2e-6, atol=1e-7)


@pytest.mark.parametrize("k", [3, 5, 7])
def test_nms_windows_exact(golden, k):
    from oracle import infer_ref as O
    from cet_pick_amd.models import decode as Dm
    from cet_pick_amd.utils import image as Im
    g = golden("decode_small.npz")
    hm = g["sigmoid"]
    t = dev(hm)[None, None]
    for fn, win in ((Dm._nms, (3, k, k)), (Dm._nms_xy, (1, k, k)), (Dm._nms_z, (k, 1, 1)), (Im._nms, (k, k, k))):
        got = fn(t, kernel=k)[0, 0].cpu().numpy()
        np.testing.assert_array_equal(got, O.nms_window(hm, win))
    if k in (3, 5):
        np.testing.assert_array_equal(Dm._nms(t, kernel=k)[0, 0].cpu().numpy(), g[f"nms_3kk_{k}"])
        np.testing.assert_array_equal(Im._nms(t, kernel=k)[0, 0].cpu().numpy(), g[f"nms_kkk_{k}"])


@pytest.mark.parametrize("shape", [(5, 17, 23), (9, 64, 64), (20, 33, 130), (3, 16, 260)])
def test_nms_ragged_shapes(shape):
    from oracle import infer_ref as O
    from cet_pick_amd.models import decode as Dm
    rng = np.random.default_rng(1)
    # coarse quantisation -> many exact ties / plateaus
    vol = np.round(rng.standard_normal(shape) * 2).astype(np.float32) / 2
    t = dev(vol)[None, None]
    for k in (3, 5):
        np.testing.assert_array_equal(Dm._nms(t, k)[0, 0].cpu().numpy(), O.nms_window(vol, (3, k, k)))
        np.testing.assert_array_equal(Dm._nms_xy(t, k)[0, 0].cpu().numpy(), O.nms_window(vol, (1, k, k)))
        np.testing.assert_array_equal(Dm._nms_z(t, k)[0, 0].cpu().numpy(), O.nms_window(vol, (k, 1, 1)))


def _cmp_dets(got, want, floor=1.5e-4):
    got = got[got[:, 3] > floor]
    want = want[want[:, 3] > floor]
    assert got.shape == want.shape
    np.testing.assert_array_equal(got[:, :3], want[:, :3])
    np.testing.assert_allclose(got[:, 3:], want[:, 3:], rtol=1e-5)


def test_tomo_decode_golden(golden):
    from cet_pick_amd.models import decode as Dm
    g = golden("decode_small.npz")
    hm = dev(g["sigmoid"])[None, None]
    for k in (3, 5):
        _cmp_dets(Dm.tomo_decode(hm, kernel=k, K=50)[0].cpu().numpy(), g[f"decode_{k}"])
        _cmp_dets(Dm.tomo_decode(hm, kernel=k, K=50, if_fiber=True)[0].cpu().numpy(), g[f"decode_fiber_{k}"])
    # fused sigmoid + decode from the raw logits
    heat, dets = Dm.sigmoid_tomo_decode(dev(g["logits"])[None, None], kernel=3, K=50)
    np.testing.assert_allclose(heat[0, 0].cpu().numpy(), g["sigmoid"], rtol=4e-6, atol=1e-7)   # fused pass: v_exp/v_rcp
    _cmp_dets(dets[0].cpu().numpy(), g["decode_3"])


def test_topk_matches_reference_indices(golden):
    from cet_pick_amd.models import decode as Dm
    g = golden("decode_small.npz")
    nms = dev(g["nms_3kk_3"])[None, None]
    s, z, y, x, inds = Dm._topk(nms, K=40)
    m = g["topk_scores"] > 1.5e-4
    np.testing.assert_array_equal(s[0, 0].cpu().numpy()[m], g["topk_scores"][m])
    np.testing.assert_array_equal(inds[0].cpu().numpy()[m], g["topk_inds"][m])
    np.testing.assert_array_equal(z[0].cpu().numpy()[m], g["topk_z"][m])
    np.testing.assert_array_equal(y[0].cpu().numpy()[m], g["topk_y"][m])
    np.testing.assert_array_equal(x[0].cpu().numpy()[m], g["topk_x"][m])


def test_decode_vs_oracle_medium():
    from oracle import infer_ref as O
    from cet_pick_amd.models import decode as Dm
    logits = make_logits((32, 96, 128), seed=3)
    heat, dets = Dm.sigmoid_tomo_decode(dev(logits)[None, None], kernel=3, K=300)
    hm = heat[0, 0].cpu().numpy()
    np.testing.assert_allclose(hm, O.sigmoid_clamp(logits), rtol=4e-6, atol=1e-7)
    want = O.tomo_decode(hm, kernel=3, K=300)      # oracle continues from the GPU's heat bits
    _cmp_dets(dets[0].cpu().numpy(), want)


def test_topk_degenerate_plateau():
    """All-equal heat: every voxel survives NMS; ties resolve to the lowest flat indices."""
    from cet_pick_amd.models import decode as Dm
    hm = torch.full((1, 1, 6, 40, 72), 0.5, device="cuda")
    dets = Dm.tomo_decode(hm, kernel=3, K=100)[0].cpu().numpy()
    idx = (dets[:, 2] * 40 + (dets[:, 1] - 0.25)) * 72 + (dets[:, 0] - 0.25)
    np.testing.assert_array_equal(idx, np.arange(100))
    assert np.all(dets[:, 3] == 0.5)


def test_topk_fewer_than_k():
    from cet_pick_amd.models import decode as Dm
    hm = torch.zeros((1, 1, 4, 16, 64), device="cuda")
    hm[0, 0, 2, 5, 7] = 0.9
    hm[0, 0, 1, 9, 60] = 0.7
    dets = Dm.tomo_decode(hm, kernel=3, K=10)[0].cpu().numpy()
    np.testing.assert_allclose(dets[0], [7.25, 5.25, 2, 0.9, 0.9], rtol=1e-6)
    np.testing.assert_allclose(dets[1], [60.25, 9.25, 1, 0.7, 0.7], rtol=1e-6)
    assert np.all(dets[2:, 3] == 0)


@pytest.mark.parametrize("sigma", [1.0, 2, 3, 5])
def test_gaussian_vs_oracle(golden, sigma):
    from oracle import infer_ref as O
    from cet_pick_amd.utils import image as Im
    g = golden("dog_small.npz")
    shape = tuple(int(v) for v in g["shape"])
    vol, _ = make_tomo(shape, seed=317)
    got = Im.gaussian_filter(vol, sigma).cpu().numpy()
    want = O.gaussian_filter(vol.astype(np.float64), sigma)
    np.testing.assert_allclose(got, want, rtol=0, atol=2e-5)
    if sigma in (2, 3, 5):
        np.testing.assert_allclose(got[18, ::3], g[f"gauss{int(sigma)}_z18"], rtol=0, atol=2e-5)
        np.testing.assert_allclose(got[0, ::3], g[f"gauss{int(sigma)}_z0"], rtol=0, atol=2e-5)


@pytest.mark.parametrize("sigma", [2.5, 5.0, 6.0])     # radius 10 / 20 (specialised kernels) / 24 (generic fallback)
@pytest.mark.parametrize("shape", [(7, 9, 11), (24, 50, 300), (40, 70, 65), (13, 21, 515)])
def test_gaussian_small_and_ragged(shape, sigma):
    from oracle import infer_ref as O
    from cet_pick_amd.utils import image as Im
    rng = np.random.default_rng(2)
    vol = rng.standard_normal(shape).astype(np.float32)
    got = Im.gaussian_filter(vol, sigma).cpu().numpy()   # radius > some dims: multi-reflection; rows % 4, W % 4 != 0
    want = O.gaussian_filter(vol.astype(np.float64), sigma)
    np.testing.assert_allclose(got, want, rtol=0, atol=2e-5)


def test_greedy_nms_golden(golden):
    from cet_pick_amd.utils import image as Im
    g = golden("greedy_small.npz")
    vol = g["vol"]
    for d, thr in ((6, 0.5), (14, 1.0), (4, -np.inf)):
        s, c = Im.non_maximum_suppression_3d(vol, d, threshold=thr)
        np.testing.assert_array_equal(c, g[f"coords_d{d}"])
        np.testing.assert_array_equal(s, g[f"scores_d{d}"])
    s, c = Im.non_maximum_suppression_3d(vol, 6, scale=1.5, threshold=0.2)
    np.testing.assert_array_equal(c, g["coords_d6_s15"])


@pytest.mark.parametrize("case", ["plateau", "many", "skewed"])
def test_greedy_nms_output_order_by_intervals_and_by_counting(case, monkeypatch):
    """emit_rank_kernel, round 5: picks ranked by score intervals (cells dealt to the workgroups by the running count) - and the n^2
    counting form it keeps for a cell of more than 2,048 equal-score picks (`plateau`: 6,859 isolated peaks of ONE value: ordered by the
    kernels' priority key (score bits, flat index), greater key first, in both forms), with more picks than a workgroup's threads
    hold in registers (`many`: 35,937 peaks > 16 x 1,024) and with scores crowded at one end (`skewed`).  Isolated peaks further apart
    than the suppression distance: every peak is a pick, so the expected output is the peaks sorted - exact; both forms
    (MI_EMIT_RANK_N2=1) give the same arrays."""
    from cet_pick_amd.utils import image as Im
    rng = np.random.default_rng({"plateau": 1, "many": 2, "skewed": 3}[case])
    step, d = 6, 4
    m = 19 if case == "plateau" else 33 if case == "many" else 24
    shape = (m * step, m * step, m * step)
    vol = np.zeros(shape, np.float32)
    zz, yy, xx = np.meshgrid(*[np.arange(m) * step + 2] * 3, indexing="ij")
    n = m ** 3
    if case == "plateau":
        vals = np.full(n, 3.0, np.float32)
    elif case == "many":
        vals = (1.0 + rng.permutation(n) / n).astype(np.float32)
    else:
        vals = (1.0 + 1e-3 * rng.random(n) ** 8).astype(np.float32)
        vals[:5] = 50.0 + np.arange(5)
        vals = np.unique(vals)                 # unique scores (ties are ordered by index: covered by `plateau`)
        vals = rng.permutation(vals)
        n = len(vals)
    flat = (zz.ravel() * shape[1] + yy.ravel()) * shape[2] + xx.ravel()
    flat = flat[:n]
    vol.ravel()[flat] = vals
    s, c = Im.non_maximum_suppression_3d(vol, d, threshold=0.5)
    assert len(s) == n
    order = np.lexsort((flat, -vals.astype(np.float64)))          # score descending, flat index ascending among equals
    if case == "plateau":
        # equal scores: the priority key is (score bits, flat index) - the greater key first
        order = np.argsort(-flat, kind="stable")
    np.testing.assert_array_equal(s, vals[order])
    want = np.stack([flat[order] % shape[2], (flat[order] // shape[2]) % shape[1], flat[order] // (shape[1] * shape[2])], 1)
    np.testing.assert_array_equal(c, want)
    monkeypatch.setenv("MI_EMIT_RANK_N2", "1")
    s2, c2 = Im.non_maximum_suppression_3d(vol, d, threshold=0.5)
    np.testing.assert_array_equal(s2, s)
    np.testing.assert_array_equal(c2, c)


def test_dog_pick_golden(golden):
    from cet_pick_amd.utils import image as Im
    g = golden("dog_small.npz")
    shape = tuple(int(v) for v in g["shape"])
    vol, _ = make_tomo(shape, seed=317)
    for sigmas in ((2, 4), (3, 5), (2, 4, 6)):
        tag = "_".join(map(str, sigmas))
        s, c = Im.get_potential_coords_pyramid(vol, sigmas=list(sigmas))
        np.testing.assert_array_equal(c, g[f"coords_{tag}"])
        np.testing.assert_allclose(s, g[f"scores_{tag}"], rtol=1e-3)


def test_dog_pick_vs_oracle_larger():
    from oracle import infer_ref as O
    from cet_pick_amd.utils import image as Im
    vol, _ = make_tomo((48, 200, 264), seed=318)
    s, c = Im.get_potential_coords_pyramid(vol, sigmas=[3, 5])
    so, co = O.get_potential_coords_pyramid(vol.astype(np.float64), sigmas=(3, 5))
    assert len(s) > 20
    # picks whose score clears the cutoff by a margin must agree exactly (fp32 vs fp64 DoG)
    heat = O.dog_nms_heat(vol.astype(np.float64), (3, 5))
    cut = O.pos_threshold(heat)
    strong_o = {tuple(r) for r, sc in zip(co, so) if sc > cut * 1.01}
    strong_g = {tuple(r) for r, sc in zip(c, s) if sc > cut * 1.01}
    assert strong_o == strong_g
    assert abs(len(s) - len(so)) <= max(2, len(so) // 100)


@pytest.mark.parametrize("shape,sigmas,nms_d", [((48, 200, 264), (3, 5), 14), ((40, 160, 512), (2, 4), 14),
                                                ((64, 256, 256), (3, 5), 30), ((44, 130, 96), (2.5, 4.5), 9)])
def test_dog_pick_chains_agree(shape, sigmas, nms_d, monkeypatch):
    """The picker's three implementations of one algorithm on the same tomogram: the round-4 chain (z + x | y + DoG + NMS,
    candidates as a grid index), the same filters with round 3's bitmap neighbour search, and round 3's chain (z | y | x +
    DoG + NMS).  Index against bitmap: the same candidates go through two neighbour searches - picks and scores must be
    IDENTICAL (nms_d = 30: balls of 33 planes, neighbour lists longer than a row of `nbr` - the rounds' re-probe runs on
    both).  Against round 3's chain the filter passes run in another order (rounding-order differences in the DoG): the
    strong picks agree exactly, scores to 1e-5."""
    from cet_pick_amd.utils import image as Im
    vol, _ = make_tomo(shape, seed=321)
    v = dev(vol)

    def run():
        s, c, n, cut, heat = Im.dog_pick(v, list(sigmas), nms_d=nms_d, return_heat=True)
        k = int(n.item())
        assert k > 0
        return s[:k].cpu().numpy(), c[:k].cpu().numpy(), float(cut.item()), heat.cpu().numpy()

    s0, c0, cut0, h0 = run()
    monkeypatch.setenv("MI_DOG_NO_INDEX", "1")
    s1, c1, cut1, h1 = run()
    monkeypatch.delenv("MI_DOG_NO_INDEX")
    assert cut0 == cut1
    np.testing.assert_array_equal(h0, h1)
    np.testing.assert_array_equal(c0, c1)
    np.testing.assert_array_equal(s0, s1)
    monkeypatch.setenv("MI_NO_DOGF", "1")
    s2, c2, cut2, h2 = run()
    monkeypatch.delenv("MI_NO_DOGF")
    assert abs(cut2 - cut0) <= 1e-5 * abs(cut0)
    np.testing.assert_allclose(h2, h0, rtol=0, atol=2e-6)
    strong0 = {tuple(r) for r, sc in zip(c0, s0) if sc > cut0 * 1.01}
    strong2 = {tuple(r) for r, sc in zip(c2, s2) if sc > cut0 * 1.01}
    assert strong0 == strong2
    # the picks are the greedy NMS of the heat-map this chain produced: the oracle's sequential loop on it
    from oracle import infer_ref as O
    so, co = O.non_maximum_suppression_3d(h0, nms_d, threshold=cut0)
    np.testing.assert_array_equal(c0, co)
    np.testing.assert_array_equal(s0, so)


def test_get_potential_coords(golden):
    from cet_pick_amd.utils import image as Im
    g = golden("dog_small.npz")
    shape = tuple(int(v) for v in g["shape"])
    vol, _ = make_tomo(shape, seed=317)
    z, y, x = Im.get_potential_coords(vol, sigma1=2, sigma2=4, kernel=3, K=64)
    got = set(zip(z[0].cpu().numpy().tolist(), y[0].cpu().numpy().tolist(), x[0].cpu().numpy().tolist()))
    want = set(zip(g["gpc_z"].tolist(), g["gpc_y"].tolist(), g["gpc_x"].tolist()))
    assert len(got & want) >= 62    # fp32 vs fp64 DoG may reorder the last near-ties


@pytest.mark.parametrize("shape", [(5, 17, 24), (9, 64, 64), (20, 33, 260), (3, 16, 512), (17, 70, 256), (23, 130, 768),
                                   (1, 4, 4), (2, 3, 8)])
def test_register_march_333_exact(shape, monkeypatch):
    """The (3,3,3) register march (infer_peak3.hip: wave strips, DPP x neighbours, per-wave candidate segments) against
    the oracle and against the LDS-tile march it replaces, on plateau-rich volumes: W <= 256 (one strip per row), wider
    rows (strip-edge halo loads), ragged H and a last strip with a single active lane."""
    from oracle import infer_ref as O
    from cet_pick_amd.models import decode as Dm
    rng = np.random.default_rng(shape[1])
    vol = np.round(rng.standard_normal(shape) * 2).astype(np.float32) / 2
    t = dev(vol)[None, None]
    want = O.nms_window(vol, (3, 3, 3))
    got = Dm._nms(t, 3)[0, 0].cpu().numpy()
    np.testing.assert_array_equal(got, want)
    monkeypatch.setenv("MI_NO_PEAK3", "1")
    np.testing.assert_array_equal(Dm._nms(t, 3)[0, 0].cpu().numpy(), want)
    monkeypatch.delenv("MI_NO_PEAK3")
    # fused decode on the same shape: positive heat with ties; K above and below the number of maxima
    heat = (np.abs(vol) / 8 + 0.05).astype(np.float32)
    for K in (7, 300):
        dets = Dm.tomo_decode(dev(heat)[None, None], kernel=3, K=K)[0].cpu().numpy()
        _cmp_dets(dets, O.tomo_decode(heat, kernel=3, K=K), floor=0)
        monkeypatch.setenv("MI_NO_PEAK3", "1")
        dets2 = Dm.tomo_decode(dev(heat)[None, None], kernel=3, K=K)[0].cpu().numpy()
        monkeypatch.delenv("MI_NO_PEAK3")
        np.testing.assert_array_equal(dets, dets2)


def test_register_march_inf_peak():
    """An infinite peak is a maximum and suppresses its window, as in torch's max_pool3d form of `_nms`."""
    from oracle import infer_ref as O
    from cet_pick_amd.models import decode as Dm
    rng = np.random.default_rng(5)
    vol = rng.standard_normal((6, 20, 64)).astype(np.float32)
    vol[2, 7, 9] = np.inf
    t = dev(vol)[None, None]
    got = Dm._nms(t, 3)[0, 0].cpu().numpy()
    want = torch.nn.functional.max_pool3d(torch.as_tensor(vol)[None, None], 3, 1, 1)
    want = (torch.as_tensor(vol)[None, None] * (want == torch.as_tensor(vol)[None, None]).float())[0, 0].numpy()
    np.testing.assert_array_equal(got, want)


def test_decode_large_k_sorts_exactly():
    """K = 5000 (the `_topk` of get_potential_coords): the single-workgroup sorter works on 8192 keys."""
    from oracle import infer_ref as O
    from cet_pick_amd.models import decode as Dm
    rng = np.random.default_rng(11)
    heat = rng.random((12, 64, 128)).astype(np.float32) * 0.9 + 0.05
    dets = Dm.tomo_decode(dev(heat)[None, None], kernel=3, K=5000)[0].cpu().numpy()
    _cmp_dets(dets, O.tomo_decode(heat, kernel=3, K=5000), floor=0)
    s = dets[:, 3][dets[:, 3] > 0]
    assert np.all(np.diff(s) <= 0)


def test_decode_workspace_stays_clean_between_calls():
    """The Python mirror passes the 'header is clean' bit: no clearing pass per decode, the final kernel's last workgroup
    zeroes the header.  Repeated and interleaved calls (different shapes, K, the plateau slow path, the generic march)
    must keep giving the oracle's answer."""
    from oracle import infer_ref as O
    from cet_pick_amd.models import decode as Dm
    rng = np.random.default_rng(21)
    a = (rng.random((10, 40, 64)) * 0.9 + 0.05).astype(np.float32)
    b = (rng.random((7, 33, 72)) * 0.9 + 0.05).astype(np.float32)
    wa, wb = O.tomo_decode(a, kernel=3, K=200), O.tomo_decode(b, kernel=5, K=60)
    plateau = torch.full((1, 1, 6, 40, 72), 0.5, device="cuda")
    for _ in range(3):
        _cmp_dets(Dm.tomo_decode(dev(a)[None, None], kernel=3, K=200)[0].cpu().numpy(), wa, floor=0)
        _cmp_dets(Dm.tomo_decode(dev(b)[None, None], kernel=5, K=60)[0].cpu().numpy(), wb, floor=0)
        d = Dm.tomo_decode(plateau, kernel=3, K=100)[0].cpu().numpy()       # > MI_SEL_CAP ties: radix-select path
        assert np.all(d[:, 3] == 0.5)
    # the C-ABI without the bit needs no initialisation and tolerates a dirty header
    from cet_pick_amd import _lib as L
    lib = L.lib()
    ws = torch.full((int(lib.mi_decode_workspace_bytes(10, 40, 64, 200)),), 0xAB, dtype=torch.uint8, device="cuda")
    dets = torch.empty((200, 5), device="cuda")
    t = dev(a)
    for _ in range(2):
        L.check(lib.mi_sigmoid_nms_topk(L.ptr(t), None, 10, 40, 64, 3, 0, 0, 200, L.ptr(dets), None, L.ptr(ws), ws.numel(),
                                        L.stream()), "mi_sigmoid_nms_topk")
        _cmp_dets(dets.cpu().numpy(), wa, floor=0)


def test_decode_one_launch_paths_match_the_chain_and_the_oracle(monkeypatch):
    """infer_decode1.hip (one launch: per-workgroup best lists + selection by the last workgroup) against the three-launch
    chain (MI_DECODE_CHAIN=1) and the oracle, on inputs that take each of its paths: the plain one; every one of the K best
    inside ONE workgroup's region (its list is too short: the bound check sends the call to the exact radix select over
    the segments); K above what the tables hold; a grid of more than 256 workgroups (table read in chunks); a plateau
    region (lanes run out of list slots: spill)."""
    from oracle import infer_ref as O
    from cet_pick_amd.models import decode as Dm

    def both(logits, K, sig=True):
        t = dev(logits)[None, None]
        fn = (lambda: Dm.sigmoid_tomo_decode(t, kernel=3, K=K)) if sig else (lambda: (t, Dm.tomo_decode(t, kernel=3, K=K)))
        heat, d1 = fn()
        monkeypatch.setenv("MI_DECODE_CHAIN", "1")
        heat2, d2 = fn()
        monkeypatch.delenv("MI_DECODE_CHAIN")
        np.testing.assert_array_equal(heat[0, 0].cpu().numpy(), heat2[0, 0].cpu().numpy())
        np.testing.assert_array_equal(d1[0].cpu().numpy(), d2[0].cpu().numpy())
        want = O.tomo_decode(heat[0, 0].cpu().numpy(), kernel=3, K=K)
        _cmp_dets(d1[0].cpu().numpy(), want, floor=0)
        return d1[0].cpu().numpy()

    rng = np.random.default_rng(5)
    # plain: the benchmark's kind of volume, smaller
    both(make_logits((32, 128, 256), seed=9), 300)
    # all strong peaks inside one workgroup's 32 rows x 4 planes (z 0..3, y 0..31): more than its list holds
    lg = (rng.standard_normal((16, 64, 256)) - 4).astype(np.float32)
    zz, yy, xx = np.meshgrid(np.arange(0, 4, 2), np.arange(1, 31, 3), np.arange(2, 250, 4), indexing="ij")
    lg[zz.ravel(), yy.ravel(), xx.ravel()] = rng.uniform(2, 6, zz.size).astype(np.float32)
    d = both(lg, 200)
    assert np.all(d[:, 2] < 4) and np.all(d[:, 1] < 32)
    # K larger than (workgroups x list length)
    both(make_logits((8, 64, 64), seed=2), 900)
    # 512 workgroups (2 x 32 x 4): the selecting workgroup reads the table in two chunks
    both(make_logits((16, 1024, 512), seed=4), 500)
    # plateau slab inside noise (every voxel of the slab is a maximum of its window): spill path, ties by index
    hm = (rng.random((12, 40, 256)) * 0.5 + 0.05).astype(np.float32)
    hm[4:8, 8:24, 64:192] = 0.9
    both(hm, 700, sig=False)
