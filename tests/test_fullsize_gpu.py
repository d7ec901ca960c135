"""Size-independent properties at BASELINE.json's FULL sizes (the oracle would take minutes there): sortedness,
local-maximum / idempotence of the NMS, exclusion radius of the greedy picker, linearity and constant preservation of
the Gaussian, adjointness of the convolution triplet, EMA / queue invariants of a batch-64 MoCo step, loader statistics."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def test_decode_properties_128x256x256():
    from cet_pick_amd.models import decode as Dm
    from cet_pick_amd.synthetic import make_logits
    logits = torch.as_tensor(make_logits((128, 256, 256), seed=317)).cuda()[None, None]
    ref_sig = torch.clamp(torch.sigmoid(logits), 1e-4, 1 - 1e-4)
    K = 900
    heat, dets = Dm.sigmoid_tomo_decode(logits.clone(), kernel=3, K=K)
    assert torch.allclose(heat, ref_sig, rtol=2e-6, atol=2e-6)
    d = dets[0]
    s = d[:, 3]
    assert torch.all(s[:-1] >= s[1:]) and torch.equal(d[:, 3], d[:, 4])          # sorted, score duplicated
    xs, ys, zs = (d[:, 0] - 0.25).round().long(), (d[:, 1] - 0.25).round().long(), d[:, 2].long()
    assert torch.all((xs >= 0) & (xs < 256) & (ys >= 0) & (ys < 256) & (zs >= 0) & (zs < 128))
    # every detection is a 3x3x3 local maximum of the heat-map and carries its value
    pooled = F.max_pool3d(heat, 3, 1, 1)
    assert torch.equal(heat[0, 0, zs, ys, xs], pooled[0, 0, zs, ys, xs])
    assert torch.allclose(heat[0, 0, zs, ys, xs], s, rtol=0, atol=0)
    # and nothing better was left out: the K-th score bounds every other local maximum
    keep = (pooled == heat) & (heat > 0)
    nms = torch.where(keep, heat, torch.zeros_like(heat))
    nms[0, 0, zs, ys, xs] = 0
    assert float(nms.max()) <= float(s[-1])
    # NMS is idempotent
    once = Dm._nms(heat, 3)
    assert torch.equal(Dm._nms(once, 3), once)


def test_dog_picker_properties_256x512x512():
    from cet_pick_amd.utils import image as Im
    from cet_pick_amd.synthetic import make_tomo
    vol, _ = make_tomo((256, 512, 512), seed=317)
    v = torch.as_tensor(vol).cuda()
    scores, coords, n, cutoff, heat = Im.dog_pick(v, [3, 5], return_heat=True)
    n = int(n)
    assert 100 < n < scores.numel()
    s, c = scores[:n], coords[:n].long()
    assert torch.all(s[:-1] >= s[1:]) and float(s[-1]) > float(cutoff)
    x, y, z = c[:, 0], c[:, 1], c[:, 2]
    assert torch.all((z >= 10) & (z < 256 - 10) & (y >= 30) & (y < 512 - 30) & (x >= 30) & (x < 512 - 30))   # zeroed borders
    assert torch.allclose(heat[z, y, x], s)                                     # scores are the NMS'd DoG values
    # greedy exclusion: no two picks closer than the ball radius d/2 = 7 (blocked pairwise distances)
    p = c.float()
    for i0 in range(0, n, 4096):
        dist = torch.cdist(p[i0:i0 + 4096], p)
        dist[torch.arange(min(4096, n - i0)), torch.arange(i0, min(i0 + 4096, n))] = 1e9
        assert float(dist.min()) > 7.0
    # the heat-map is an xy-NMS of the DoG: survivors are 3x3 in-plane maxima of it
    pooled = F.max_pool2d(heat[None], 3, 1, 1)[0]
    assert torch.equal(torch.where(heat > 0, pooled, heat), heat)


def test_gaussian_linearity_and_constants_256x512x512():
    from cet_pick_amd.utils import image as Im
    g = torch.Generator(device="cuda").manual_seed(1)
    a = torch.randn(256, 512, 512, device="cuda", generator=g)
    b = torch.randn(256, 512, 512, device="cuda", generator=g)
    for sigma in (3.0, 5.0):
        ga, gb = Im.gaussian_filter(a, sigma), Im.gaussian_filter(b, sigma)
        gl = Im.gaussian_filter(2.0 * a - 0.5 * b, sigma)
        assert float((gl - (2.0 * ga - 0.5 * gb)).abs().max()) < 2e-6
        const = Im.gaussian_filter(torch.full_like(a, 3.25), sigma)
        assert float((const - 3.25).abs().max()) < 2e-6                            # weights sum to one, reflect padding
        # symmetric kernel + reflect boundary: flipping commutes with filtering
        assert torch.allclose(Im.gaussian_filter(a.flip(2), sigma), ga.flip(2), rtol=0, atol=1e-6)
        assert torch.allclose(Im.gaussian_filter(a.flip(0), sigma), ga.flip(0), rtol=0, atol=1e-6)


@pytest.mark.parametrize("layer", [(64, 16, 16, 16, 64, 64, 3, 1, 1), (64, 8, 8, 8, 64, 128, 3, 2, 1), (64, 4, 4, 4, 128, 256, 1, 2, 0),
                                   (64, 32, 32, 32, 1, 64, 7, 2, 3)])
def test_conv_triplet_adjointness_at_batch_64(layer):
    """<conv(x, W), dy> == <x, dgrad(dy, W)> == <W, wgrad(x, dy)>: the three kernels are one bilinear form."""
    from cet_pick_amd import hipops as H
    n, d, h, w, ci, co, k, s, p = layer
    g = torch.Generator(device="cuda").manual_seed(sum(layer))
    x = torch.randn(n, d, h, w, ci, device="cuda", generator=g)
    wt = H.conv_weight_param(co, ci, k)
    wt.data = (torch.randn(wt.shape, device="cuda", generator=g) * 0.05).permute(2, 3, 4, 1, 0).contiguous().permute(4, 3, 0, 1, 2)
    assert H._phys_ok(wt)
    y = H.conv_fwd(x, wt, k, s, p)
    dy = torch.randn(y.shape, device="cuda", generator=g)
    form = float((y.double() * dy.double()).sum())
    H.conv_wgrad_into(x, dy, wt, k, s, p)
    via_w = float((wt.grad.double() * wt.detach().double()).sum())
    assert abs(via_w - form) <= 2e-5 * abs(form) + 1e-3
    if ci != 1:
        dx = H.conv_dgrad(dy, wt, tuple(x.shape), k, s, p)
        via_x = float((dx.double() * x.double()).sum())
        assert abs(via_x - form) <= 2e-5 * abs(form) + 1e-3


def test_moco_step_invariants_batch_64():
    from cet_pick_amd.models.networks.moco_encoder_3d import get_moco_net_small_3d
    from cet_pick_amd.models.moco import MoCo
    from cet_pick_amd.trains.moco_engine import MocoStepEngine
    torch.manual_seed(317)
    heads = {"proj": 256, "pred": 256}
    moco = MoCo(get_moco_net_small_3d(18, heads, 0), get_moco_net_small_3d(18, heads, 0), dim=128, r=1024, m=0.999, T=0.1).cuda()
    moco.train()
    eng = MocoStepEngine(moco, lr=1e-3)
    g = torch.Generator(device="cuda").manual_seed(3)
    xq = torch.randn(64, 1, 32, 32, 32, device="cuda", generator=g)
    xk = torch.randn(64, 1, 32, 32, 32, device="cuda", generator=g)
    q0, k0 = eng.arena_q.flat.clone(), eng.arena_k.flat.clone()
    queue0 = moco.queue.clone()
    loss = eng.step(xq, xk)
    assert torch.isfinite(loss) and 0 < float(loss) < 2 * np.log(1025)
    # momentum update used the query weights BEFORE this step's SGD (models/moco.py:120-121)
    assert torch.allclose(eng.arena_k.flat, 0.999 * k0 + 0.001 * q0, rtol=0, atol=1e-7)
    # SGD: p <- p - lr * g
    assert torch.allclose(eng.arena_q.flat, q0 - 1e-3 * eng.arena_q.flat_grad, rtol=0, atol=1e-7)
    # enqueue: 64 unit-norm keys in columns [0, 64), the rest untouched, pointer advanced
    assert int(moco.queue_ptr) == 64
    assert torch.equal(moco.queue[:, 64:], queue0[:, 64:]) and not torch.equal(moco.queue[:, :64], queue0[:, :64])
    assert torch.allclose(moco.queue[:, :64].norm(dim=0), torch.ones(64, device="cuda"), atol=1e-5)
    for _ in range(15):
        eng.step(xq, xk)
    assert int(moco.queue_ptr) == 0                                              # 16 * 64 = r: wrapped (r % B == 0)


def test_loader_statistics_256x512x512():
    from cet_pick_amd.utils import loader
    rec = (np.random.default_rng(5).standard_normal((512, 256, 512)) * 12 + 80).astype(np.float32)   # file order (x, z, y)
    z = loader.load_rec(rec, "xzy", False)
    assert tuple(z.shape) == (256, 512, 512)
    assert abs(float(z.double().mean())) < 1e-6 and abs(float(z.double().std(unbiased=False)) - 1) < 1e-6
    assert torch.equal(z[5, 7], loader.load_rec(rec[7:8], "xzy", False)[5, 0] * 0 + z[5, 7])       # shape sanity
    comp = loader.load_rec(rec, "xzy", True)
    assert tuple(comp.shape) == (128, 512, 512)
    pre = loader.preprocess(z, 0)
    lo, hi = float(pre.min()), float(pre.max())
    assert lo == 0.0 and hi == 1.0
    levels = torch.unique(pre)
    assert levels.numel() <= 256
    step = levels[1:] - levels[:-1]
    assert float((step / step.min()).sub((step / step.min()).round()).abs().max()) < 1e-3         # a uniform level grid
