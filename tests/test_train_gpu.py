"""GPU parity of the training path (through the C-ABI) against the torch-CPU oracle and the
golden vectors generated from the reference.  fp32 tolerance 1e-3 (north_star), usually far
tighter because the f32 MFMA is an exact fmaf chain."""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def cl(x):  # NCDHW -> channels-last contiguous on the GPU
    return x.permute(0, 2, 3, 4, 1).contiguous().cuda()


def ncdhw(x):
    return x.permute(0, 4, 1, 2, 3).contiguous().cpu()


def make_w(co, ci, k, g):
    from cet_pick_amd import hipops as H
    p = H.conv_weight_param(co, ci, k)
    w = torch.randn(co, ci, k, k, k, generator=g) * (2.0 / (ci * k ** 3)) ** 0.5
    with torch.no_grad():
        p.copy_(w)
    p.data = p.data.cuda()
    return p, w


CASES = [
    # N, D, H, W, Ci, Co, k, stride, pad
    (2, 8, 8, 8, 64, 64, 3, 1, 1),
    (2, 8, 8, 8, 64, 128, 3, 2, 1),
    (3, 4, 4, 4, 128, 128, 3, 1, 1),
    (2, 8, 8, 8, 64, 128, 1, 2, 0),
    (4, 2, 2, 2, 256, 256, 3, 1, 1),
    (1, 5, 6, 7, 16, 32, 3, 1, 1),       # non power-of-two grid
    (2, 7, 5, 6, 32, 16, 3, 2, 1),
    (2, 16, 16, 16, 1, 64, 7, 2, 3),     # stem
    (64, 1, 1, 1, 256, 128, 1, 1, 0),    # linear
    (70, 4, 4, 4, 64, 128, 3, 2, 1),     # 2^3 output, stride 2, two sample chunks: pair_wgrad_kernel
    (5, 2, 2, 2, 128, 64, 3, 1, 1),
    (70, 4, 4, 4, 128, 128, 3, 1, 1),    # layer2, two sample chunks: pair_wgrad_kernel in segments (round 4)
    (5, 6, 6, 6, 64, 64, 3, 2, 1),       # 3^3 output
]


@pytest.mark.parametrize("arith", ["bf16x3", "f32"])
@pytest.mark.parametrize("case", CASES)
def test_conv_fwd_dgrad_wgrad(case, arith, monkeypatch):
    from cet_pick_amd import hipops as H
    monkeypatch.setenv("MI_CONV_ARITH", arith)      # generic kernel: bf16x3 cut (default) / native f32 MFMA
    n, d, h, w_, ci, co, k, s, p = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(n, ci, d, h, w_, generator=g)
    param, w = make_w(co, ci, k, g)
    assert H._phys_ok(param)
    y = H.conv_fwd(cl(x), param, k, s, p)
    y_ref = F.conv3d(x, w, stride=s, padding=p)
    np.testing.assert_allclose(ncdhw(y).numpy(), y_ref.numpy(), rtol=1e-4, atol=1e-4)
    # fused epilogue: residual + relu
    res = torch.randn(y_ref.shape, generator=g)
    y2 = H.conv_fwd(cl(x), param, k, s, p, cl(res), True)
    np.testing.assert_allclose(ncdhw(y2).numpy(), F.relu(y_ref + res).numpy(), rtol=1e-4, atol=1e-4)
    dy = torch.randn(y_ref.shape, generator=g)
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    F.conv3d(xr, wr, stride=s, padding=p).backward(dy)
    if ci != 1:
        dx = H.conv_dgrad(cl(dy), param, (n, d, h, w_, ci), k, s, p)
        np.testing.assert_allclose(ncdhw(dx).numpy(), xr.grad.numpy(), rtol=1e-4, atol=2e-4)
        mask = torch.randn(x.shape, generator=g)
        r2 = torch.randn(x.shape, generator=g)
        dx2 = H.conv_dgrad(cl(dy), param, (n, d, h, w_, ci), k, s, p, cl(r2), cl(mask))
        np.testing.assert_allclose(ncdhw(dx2).numpy(), ((xr.grad + r2) * (mask > 0)).numpy(), rtol=1e-4, atol=2e-4)
    param.grad = None
    H.conv_wgrad_into(cl(x), cl(dy), param, k, s, p)
    scale = float(wr.grad.abs().max())
    np.testing.assert_allclose(param.grad.cpu().numpy(), wr.grad.numpy(), rtol=1e-4, atol=2e-5 * max(scale, 1.0))
    # second call accumulates (AccumulateGrad contract)
    H.conv_wgrad_into(cl(x), cl(dy), param, k, s, p)
    np.testing.assert_allclose(param.grad.cpu().numpy(), 2 * wr.grad.numpy(), rtol=1e-4, atol=4e-5 * max(scale, 1.0))


@pytest.mark.parametrize("case", [(16, 8, 8, 8, 64, 64, 3, 1, 1), (8, 4, 4, 4, 128, 256, 3, 2, 1), (5, 4, 4, 4, 48, 16, 3, 1, 1),
                                  (16, 2, 2, 2, 256, 512, 1, 2, 0)])
def test_conv_bf16x3_is_f32_equivalent(case, monkeypatch):
    """The default arithmetic (three-way bf16 cut, six products, f32 accumulate on the bf16 matrix pipe) against float64:
    its error may not exceed the error of the f32 MFMA path (an fmaf chain) by more than rounding-order noise."""
    from cet_pick_amd import hipops as H
    n, d, h, w_, ci, co, k, s, p = case
    g = torch.Generator().manual_seed(7 + sum(case))
    # wide dynamic range: the cut must hold for every exponent, not only for unit-scale data
    x = torch.randn(n, ci, d, h, w_, generator=g) * torch.exp(3 * torch.randn(n, ci, d, h, w_, generator=g))
    param, w = make_w(co, ci, k, g)
    x64 = x.double().cuda().requires_grad_(True)
    w64 = w.double().cuda().requires_grad_(True)
    y64 = F.conv3d(x64, w64, stride=s, padding=p)
    dy = torch.randn(y64.shape, generator=g)
    gx, gw = torch.autograd.grad(y64, (x64, w64), dy.double().cuda())
    ref = (y64.detach().permute(0, 2, 3, 4, 1), gx.permute(0, 2, 3, 4, 1), gw)
    errs = {}
    for arith in ("f32", "bf16x3"):
        monkeypatch.setenv("MI_CONV_ARITH", arith)
        y = H.conv_fwd(cl(x), param, k, s, p)
        dx = H.conv_dgrad(cl(dy), param, (n, d, h, w_, ci), k, s, p)
        param.grad = None
        H.conv_wgrad_into(cl(x), cl(dy), param, k, s, p)
        outs = (y, dx, param.grad.clone())
        errs[arith] = [float((a.double() - b).abs().max() / b.abs().max()) for a, b in zip(outs, ref)]
    for e32, e3 in zip(errs["f32"], errs["bf16x3"]):
        assert e3 < 2e-6, errs                  # f32-level accuracy in absolute terms ...
        assert e3 <= 2.0 * e32 + 2e-7, errs     # ... and no worse than the f32 matrix instruction


@pytest.mark.parametrize("n,d,hw,c", [(3, 8, 8, 64), (5, 2, 8, 64), (2, 6, 8, 64), (64, 8, 8, 64),
                                      (5, 4, 4, 128), (2, 4, 4, 128), (64, 4, 4, 128),
                                      (64, 2, 2, 256), (5, 2, 2, 256), (70, 2, 2, 128),
                                      (16, 8, 8, 128), (66, 2, 8, 128), (17, 8, 8, 128),
                                      (8, 16, 16, 64), (8, 4, 24, 64), (33, 2, 16, 64),
                                      (16, 4, 4, 256), (19, 4, 4, 256), (6, 4, (16, 24), 64),
                                      (8, 8, (14, 15), 64), (4, 16, (15, 23), 64)])
def test_conv_direct3_matches_igemm_and_float64(n, d, hw, c, monkeypatch):
    """(round 4: also 128 channels on 8 x 8 planes and 64 channels on 16 x 16 / 24 x 24 planes - layer2 / layer1 of larger crops;
    round 5: ragged planes, 14 x 15 and 15 x 23 - the last tile of a row / column hangs over the edge.)
    layer1- (3^3, stride 1, 64 -> 64, 8 x 8 planes) and layer2-shaped (128 -> 128, 4 x 4 x 4; odd batch: a half-empty
    sample pair) convolutions take the patch-resident direct kernels (conv_direct3.hip), 2 x 2 x 2 volumes (layer3,
    feature_3d) the register-staged dense GEMM (conv_cube2.hip); MI_CONV_NO_DIRECT=1 keeps the implicit GEMM.  Both are the bf16x3 arithmetic: equal up to the
    summation order, and both at f32 level against float64 - forward with residual + ReLU, data gradient with
    residual + mask, weight gradient; every z tile position (first / interior / last plane pair)."""
    from cet_pick_amd import hipops as H
    from conftest import f32_equivalent
    hh, ww = hw if isinstance(hw, tuple) else (hw, hw)
    g = torch.Generator().manual_seed(100 * n + d + c)
    x = torch.randn(n, c, d, hh, ww, generator=g) * torch.exp(2 * torch.randn(n, c, d, hh, ww, generator=g))
    param, w = make_w(c, c, 3, g)
    res = torch.randn(n, c, d, hh, ww, generator=g)
    mask = torch.randn(n, c, d, hh, ww, generator=g)
    dy = torch.randn(n, c, d, hh, ww, generator=g)
    def chain(xx, ww, rr, mm, dd):
        xx = xx.clone().requires_grad_(True)
        ww = ww.clone().requires_grad_(True)
        y = F.conv3d(xx, ww, padding=1)
        gx, gw = torch.autograd.grad(y, (xx, ww), dd)
        return (F.relu(y.detach() + rr).permute(0, 2, 3, 4, 1), ((gx + rr) * (mm > 0)).permute(0, 2, 3, 4, 1), gw)
    ref64 = chain(x.double(), w.double(), res.double(), mask.double(), dy.double())
    cpu32 = chain(x, w, res, mask, dy)                   # the arbiter: the same chain in fp32 on the CPU
    out = {}
    for tag, off in (("direct", "0"), ("igemm", "1")):
        monkeypatch.setenv("MI_CONV_NO_DIRECT", off)
        yf = H.conv_fwd(cl(x), param, 3, 1, 1, cl(res), True)
        yd = H.conv_dgrad(cl(dy), param, (n, d, hh, ww, c), 3, 1, 1, cl(res), cl(mask))
        param.grad = None
        H.conv_wgrad_into(cl(x), cl(dy), param, 3, 1, 1)
        out[tag] = (yf.cpu(), yd.cpu(), param.grad.detach().cpu().clone())
    for i in range(3):
        f32_equivalent(out["direct"][i].numpy(), cpu32[i].numpy(), ref64[i].numpy(), what="direct3 %s" % ("fwd", "dgrad", "wgrad")[i])
        scale = float(ref64[i].abs().max())
        # two fp32 evaluations in different summation orders over K = 27 c terms: the bound of the 64-channel cases, scaled by sqrt(K)
        assert float((out["direct"][i] - out["igemm"][i]).abs().max()) / scale < 2e-6 * (c / 64.0) ** 0.5


@pytest.mark.parametrize("n,gi,ci,co,with_ds", [(3, 8, 64, 128, True), (64, 8, 64, 128, True), (2, 8, 64, 128, False),
                                                  (5, 4, 128, 256, True), (70, 4, 128, 256, True), (4, 4, 128, 256, False),
                                                  (3, 8, 128, 256, True), (33, 8, 128, 256, True), (2, 8, 128, 256, False)])
def test_s2_block_dgrad_matches_generic_and_float64(n, gi, ci, co, with_ds):
    """csrc/conv_s2.hip: the data gradient of a BasicBlock's stride-2 front (3^3 stride-2 convolution + 1x1 stride-2 shortcut,
    ReLU mask of the input) in one launch, against the two generic launches and against torch in float64; ragged batches
    (layer3's workgroups hold four samples), the variant without a shortcut, and layer3.0 of the 64^3 crops (8^3 -> 4^3 at 128 -> 256)."""
    from cet_pick_amd import hipops as H
    from conftest import f32_equivalent
    g = torch.Generator().manual_seed(n + gi + co)
    go = gi // 2
    param, w = make_w(co, ci, 3, g)
    pds, wds = make_w(co, ci, 1, g)
    dh = torch.randn(n, co, go, go, go, generator=g) * torch.exp(torch.randn(n, co, go, go, go, generator=g))
    d2 = torch.randn(n, co, go, go, go, generator=g)
    mask = torch.randn(n, ci, gi, gi, gi, generator=g)
    res = torch.randn(n, ci, gi, gi, gi, generator=g)
    def ref(dt):
        x = torch.zeros(n, ci, gi, gi, gi, dtype=dt, requires_grad=True)
        y = F.conv3d(x, w.to(dt), stride=2, padding=1)
        gx = torch.autograd.grad(y, x, dh.to(dt))[0]
        if with_ds:
            x2 = torch.zeros(n, ci, gi, gi, gi, dtype=dt, requires_grad=True)
            gx = gx + torch.autograd.grad(F.conv3d(x2, wds.to(dt), stride=2), x2, d2.to(dt))[0]
        return ((gx + res.to(dt)) * (mask > 0)).permute(0, 2, 3, 4, 1)
    r64, r32 = ref(torch.float64), ref(torch.float32)
    got = H.conv_dgrad_s2_block(cl(dh), cl(d2) if with_ds else None, param, pds if with_ds else None, (n, gi, gi, gi, ci), cl(res), cl(mask))
    assert got is not None
    f32_equivalent(got.cpu().numpy(), r32.numpy(), r64.numpy(), what="s2 block dgrad")
    # the two generic launches
    dres = H.conv_dgrad(cl(d2), pds, (n, gi, gi, gi, ci), 1, 2, 0, cl(res)) if with_ds else cl(res)
    gen = H.conv_dgrad(cl(dh), param, (n, gi, gi, gi, ci), 3, 2, 1, dres, cl(mask))
    assert float((got - gen).abs().max()) / float(r64.abs().max()) < 2e-6


@pytest.mark.parametrize("n,gi,ci,co", [(3, 8, 64, 128), (64, 8, 64, 128), (5, 4, 128, 256), (70, 4, 128, 256), (1, 4, 128, 256),
                                        (3, 8, 128, 256), (33, 8, 128, 256)])
def test_s2_block_fwd_matches_generic_and_float64(n, gi, ci, co, monkeypatch):
    """csrc/conv_s2.hip s2_fwd_kernel: relu(3^3 stride-2 convolution) and the 1x1 stride-2 shortcut of a BasicBlock's front in one
    launch, against the generic launches and against torch in float64; ragged batches (layer3's workgroups hold four samples);
    both workgroup shapes of the layer2 variant."""
    from cet_pick_amd import hipops as H
    from conftest import f32_equivalent
    g = torch.Generator().manual_seed(n + gi + co)
    param, w = make_w(co, ci, 3, g)
    pds, wds = make_w(co, ci, 1, g)
    x = torch.randn(n, ci, gi, gi, gi, generator=g) * torch.exp(torch.randn(n, ci, gi, gi, gi, generator=g))
    def ref(dt):
        return (F.relu(F.conv3d(x.to(dt), w.to(dt), stride=2, padding=1)).permute(0, 2, 3, 4, 1),
                F.conv3d(x.to(dt), wds.to(dt), stride=2).permute(0, 2, 3, 4, 1))
    r64, r32 = ref(torch.float64), ref(torch.float32)
    gen = (H.conv_fwd(cl(x), param, 3, 2, 1, None, True), H.conv_fwd(cl(x), pds, 1, 2, 0))
    for narrow in ("0", "1"):
        monkeypatch.setenv("MI_S2FWD_NARROW", narrow)
        got = H.conv_fwd_s2_block(cl(x), param, pds)
        assert got is not None
        for i, what in enumerate(("s2 block fwd", "s2 block shortcut")):
            f32_equivalent(got[i].cpu().numpy(), r32[i].numpy(), r64[i].numpy(), what=what)
            assert float((got[i] - gen[i]).abs().max()) / float(r64[i].abs().max()) < 2e-6
    # a shape that is not the encoder's is declined
    assert H.conv_fwd_s2_block(cl(x)[:, :, :, :, : ci // 2].contiguous(), param, pds) is None


@pytest.mark.parametrize("shape", [(4, 16, 16, 16, 64), (8, 2, 2, 2, 256), (64, 128), (6, 3, 5, 7, 32)])
@pytest.mark.parametrize("relu", [False, True])
def test_batchnorm_train_fwd_bwd(shape, relu):
    from cet_pick_amd import hipops as H
    g = torch.Generator().manual_seed(len(shape) + int(relu))
    c = shape[-1]
    x = torch.randn(shape, generator=g) * 2 + 0.5
    bn = H.HipBatchNorm(c).cuda()
    with torch.no_grad():
        bn.weight.copy_(torch.rand(c, generator=g) + 0.5)
        bn.bias.copy_(torch.randn(c, generator=g) * 0.1)
    ref = torch.nn.BatchNorm1d(c)
    with torch.no_grad():
        ref.weight.copy_(bn.weight.cpu()); ref.bias.copy_(bn.bias.cpu())
    xg = x.cuda().requires_grad_(True)
    y = bn(xg, relu=relu)
    xr = x.reshape(-1, c).clone().requires_grad_(True)
    yr = ref(xr)
    if relu:
        yr = F.relu(yr)
    np.testing.assert_allclose(y.detach().cpu().reshape(-1, c).numpy(), yr.detach().numpy(), rtol=1e-4, atol=1e-5)
    dy = torch.randn(shape, generator=g)
    y.backward(dy.cuda())
    yr.backward(dy.reshape(-1, c))
    np.testing.assert_allclose(xg.grad.cpu().reshape(-1, c).numpy(), xr.grad.numpy(), rtol=1e-3, atol=1e-5)
    np.testing.assert_allclose(bn.weight.grad.cpu().numpy(), ref.weight.grad.numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(bn.bias.grad.cpu().numpy(), ref.bias.grad.numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(bn.running_mean.cpu().numpy(), ref.running_mean.numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(bn.running_var.cpu().numpy(), ref.running_var.numpy(), rtol=1e-5, atol=1e-6)
    assert int(bn.num_batches_tracked) == 1
    bn.eval(); ref.eval()
    ye = bn(x.cuda(), relu=relu)
    yre = F.relu(ref(x.reshape(-1, c))) if relu else ref(x.reshape(-1, c))
    np.testing.assert_allclose(ye.detach().cpu().reshape(-1, c).numpy(), yre.detach().numpy(), rtol=1e-4, atol=1e-5)


def test_maxpool_avgpool():
    from cet_pick_amd import hipops as H
    g = torch.Generator().manual_seed(4)
    x = F.relu(torch.randn(3, 64, 16, 12, 10, generator=g))     # ReLU'd input: many exact-zero ties
    xg = cl(x).requires_grad_(True)
    y = H.maxpool3d(xg, 3, 2, 1)
    xr = x.clone().requires_grad_(True)
    yr = F.max_pool3d(xr, 3, stride=2, padding=1)
    np.testing.assert_array_equal(ncdhw(y.detach()).numpy(), yr.detach().numpy())
    dy = torch.randn(yr.shape, generator=g)
    y.backward(cl(dy))
    yr.backward(dy)
    np.testing.assert_allclose(ncdhw(xg.grad).numpy(), xr.grad.numpy(), rtol=1e-6, atol=1e-6)
    a = cl(x).requires_grad_(True)
    p = H.global_avgpool(a)
    np.testing.assert_allclose(p.detach().cpu().numpy(), x.mean(dim=(2, 3, 4)).numpy(), rtol=1e-5, atol=1e-6)
    p.backward(torch.ones_like(p))
    np.testing.assert_allclose(a.grad.cpu().numpy(), np.full(a.shape, 1.0 / (16 * 12 * 10), np.float32), rtol=1e-6)


def test_head_kernels():
    from cet_pick_amd import hipops as H
    g = torch.Generator().manual_seed(8)
    B, C, R, T = 16, 128, 256, 0.1
    q0 = torch.randn(B, C, generator=g); k0 = torch.randn(B, C, generator=g)
    queue = F.normalize(torch.randn(C, R, generator=g), dim=0)
    qg = q0.cuda().requires_grad_(True)
    qn = H.l2_normalize(qg)
    kn = H.l2_normalize(k0.cuda())
    logits = H.moco_logits(qn, kn, queue.cuda(), T)
    loss = H.cross_entropy_label0(logits)
    loss.backward()
    qr = q0.clone().requires_grad_(True)
    qrn = F.normalize(qr, dim=1); krn = F.normalize(k0, dim=1)
    lr = torch.cat([(qrn * krn).sum(1, keepdim=True), qrn @ queue], 1) / T
    lossr = F.cross_entropy(lr, torch.zeros(B, dtype=torch.long))
    lossr.backward()
    np.testing.assert_allclose(logits.detach().cpu().numpy(), lr.detach().numpy(), rtol=1e-4, atol=1e-4)
    assert abs(float(loss.detach()) - float(lossr.detach())) < 1e-5
    np.testing.assert_allclose(qg.grad.cpu().numpy(), qr.grad.numpy(), rtol=1e-3, atol=1e-6)


def test_ema_sgd_enqueue():
    from cet_pick_amd import hipops as H
    g = torch.Generator().manual_seed(9)
    n = 1000 * 4 + 3
    q = torch.randn(n + 1, generator=g)[:n].clone(); k = torch.randn(n, generator=g); gr = torch.randn(n, generator=g)
    qd, kd, gd = q.cuda(), k.cuda(), gr.cuda()
    H.ema_update_(kd, qd, 0.99)
    np.testing.assert_allclose(kd.cpu().numpy(), (k * 0.99 + q * (1.0 - 0.99)).numpy(), rtol=1e-6, atol=1e-7)
    H.sgd_step_(qd, gd, 0.05)
    np.testing.assert_allclose(qd.cpu().numpy(), (q - 0.05 * gr).numpy(), rtol=1e-6, atol=1e-7)
    # grad_scale (the 1 / world of the data-parallel step: the gradient arena holds the SUM over the ranks) and weight decay,
    # on a length that exercises the 16-byte body AND the scalar tail, with the learning rate read from the device
    for m in (n, 4 * 257, 5):
        p0, g0 = torch.randn(m, generator=g), torch.randn(m, generator=g)
        pd, gd2 = p0.cuda(), g0.cuda()
        H.sgd_step_(pd, gd2, 0.05, weight_decay=1e-2, grad_scale=0.5)
        want = p0.double() - 0.05 * (0.5 * g0.double() + 1e-2 * p0.double())
        np.testing.assert_allclose(pd.cpu().numpy(), want.numpy(), rtol=1e-6, atol=1e-7)
        assert torch.equal(gd2.cpu(), g0)                      # the arena keeps the sum: only the step scales it
        pd = p0.cuda()
        lr_dev = torch.tensor([0.02], dtype=torch.float32).cuda()
        H.sgd_step_(pd, gd2, 123.0, lr_dev=lr_dev, grad_scale=0.125)      # (a device lr overrides the host value)
        np.testing.assert_allclose(pd.cpu().numpy(), (p0.double() - 0.02 * 0.125 * g0.double()).numpy(), rtol=1e-6, atol=1e-7)
    queue = torch.zeros(8, 12).cuda(); ptr = torch.zeros(1, dtype=torch.long).cuda()
    ref = torch.zeros(8, 12); p = 0
    for it in range(5):
        keys = torch.randn(4, 8, generator=g)
        H.queue_enqueue_(queue, ptr, keys.cuda())
        ref[:, p:p + 4] = keys.T; p = (p + 4) % 12
        np.testing.assert_array_equal(queue.cpu().numpy(), ref.numpy())
        assert int(ptr) == p
    with pytest.raises(AssertionError):
        H.queue_enqueue_(queue, ptr, torch.randn(5, 8).cuda())


def _seeded_encoder():
    from cet_pick_amd.models.networks.moco_encoder_3d import TomoResClassifier3D, BasicBlock
    from cet_pick_amd.synthetic import seeded_state_dict
    enc = TomoResClassifier3D(BasicBlock, [2, 2, 2, 2], {"proj": 256, "pred": 256}, 0)
    enc.load_state_dict(seeded_state_dict(enc, seed=317))
    return enc.cuda()


def test_encoder_matches_reference_golden(golden):
    g = golden("enc3d.npz")
    enc = _seeded_encoder()
    x = torch.randn(4, 1, 32, 32, 32, generator=torch.Generator().manual_seed(99))
    enc.train()
    out = enc(x.cuda())[0]["proj"]
    np.testing.assert_allclose(out.detach().cpu().numpy(), g["proj_train"], rtol=1e-3, atol=1e-3)
    loss = (out * torch.linspace(-1, 1, 128).cuda()[None]).sum() + (out ** 2).sum() * 0.1
    loss.backward()
    idx = g["sample_idx"]
    for n, p in enc.named_parameters():
        if f"grad_{n}_norm" not in g.files:
            continue
        assert p.grad is not None, n
        # logical (reference) element order
        gf = p.grad.detach().cpu().contiguous().reshape(-1).numpy()
        ref_norm = float(g[f"grad_{n}_norm"])
        assert abs(np.linalg.norm(gf.astype(np.float64)) - ref_norm) <= 2e-3 * ref_norm + 1e-6, n
        ref_s = g[f"grad_{n}_sample"]
        # fc.bias feeds Linear(no bias)+BatchNorm: its true gradient is 0 and both sides hold ~1e-7 noise
        np.testing.assert_allclose(gf[idx % gf.size], ref_s, rtol=2e-3, atol=2e-3 * float(np.abs(ref_s).max()) + 2e-6,
                                   err_msg=n)
    np.testing.assert_allclose(enc.bn1.running_mean.cpu().numpy(), g["bn1_running_mean"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(enc.bn1.running_var.cpu().numpy(), g["bn1_running_var"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(enc.proj[7].running_var.cpu().numpy(), g["proj7_running_var"], rtol=1e-3, atol=1e-5)
    enc2 = _seeded_encoder()
    enc2.eval()
    ev = enc2.forward_test(x.cuda())["proj"]
    np.testing.assert_allclose(ev.cpu().numpy(), g["proj_eval"], rtol=1e-3, atol=1e-3)


@pytest.mark.parametrize("crop,batch", [(48, 3), (64, 2), (40, 2)])
def test_encoder_other_crop_sizes_match_oracle(crop, batch):
    """The reference encoder is shape-agnostic (moco_encoder_3d.py:156-236); the patch-resident kernels are built for the 32^3
    crops of the benchmark.  Other crop sizes take the stem kernel where its tiling divides the volume and the implicit GEMM
    elsewhere: forward and every parameter gradient against the CPU oracle, float64-arbitrated (VERDICT r2 item 8's test half).
    Round 4: branch-matched like the step tests - both oracle evaluations take the GPU's ReLU decisions (conftest.gpu_relu_decisions),
    after assert_relu_flips_on_edge has shown that every decision that differs from float64's own sits on a unit fp32 cannot
    resolve.  (Un-matched, ONE such unit decided whether this test passed: the stem's statistics summed in another f32 grouping -
    same convolution output bit for bit - moved crop 48's layer2.1.conv1 gradient from 8e-4 to 3e-3 of float64.)"""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
    from oracle import train_ref as T
    from conftest import f32_equivalent, gpu_relu_decisions, assert_relu_flips_on_edge
    enc = _seeded_encoder()
    enc.train()
    sd = {k: v.detach().cpu().clone() for k, v in enc.state_dict().items()}
    gen = torch.Generator().manual_seed(crop)
    x = torch.randn(batch, 1, crop, crop, crop, generator=gen)
    wv = torch.linspace(-1, 1, 128)
    masks = gpu_relu_decisions(enc, x.cuda())            # the decisions of the forward pass below (buffers restored)
    def run_ref(dt, relu_masks=None, pre=None):
        sdr = {k: (v.to(dt).clone().requires_grad_(k.endswith((".weight", ".bias"))) if v.is_floating_point() else v.clone())
               for k, v in sd.items()}
        out = T.encoder_forward(sdr, x.to(dt), train=True, pre=pre, relu_masks=relu_masks)
        loss = (out * wv.to(dt)[None]).sum() + (out ** 2).sum() * 0.1
        loss.backward()
        return out.detach(), {k: v.grad for k, v in sdr.items() if isinstance(v, torch.Tensor) and v.requires_grad and v.grad is not None}
    pre32, pre64 = {}, {}
    run_ref(torch.float32, pre=pre32)
    run_ref(torch.float64, pre=pre64)                    # float64's own branch: where does the GPU's differ?
    assert_relu_flips_on_edge(masks, pre64, pre32)
    o32, g32 = run_ref(torch.float32, relu_masks=masks)
    o64, g64 = run_ref(torch.float64, relu_masks=masks)
    out = enc(x.cuda())[0]["proj"]
    loss = (out * wv.cuda()[None]).sum() + (out ** 2).sum() * 0.1
    loss.backward()
    f32_equivalent(out.detach().cpu().numpy(), o32.numpy(), o64.numpy(), floor=2e-5, what="proj at crop %d" % crop)
    gscale = float(sum(float(v.norm()) ** 2 for v in g64.values()) ** 0.5)
    checked = 0
    for n, p in enc.named_parameters():
        if n not in g64 or n == "fc.bias" or n.startswith("pred."):
            continue
        floor = 5e-5 * gscale / (float(g64[n].norm()) + 1e-30) + 2e-5
        f32_equivalent(p.grad.detach().cpu().contiguous().numpy(), g32[n].numpy(), g64[n].numpy(), floor=floor, factor=3.0,
                       what="crop %d grad %s" % (crop, n))
        checked += 1
    assert checked >= 25


def test_state_dict_roundtrip_with_reference_layout(tmp_path):
    """Checkpoints keep the reference's keys and logical shapes; values survive save -> load."""
    enc = _seeded_encoder()
    keys = json.load(open(os.path.join(HERE, "golden", "ckpt_keys.json")))["moco3d_encoder"]
    sd = enc.state_dict()
    assert list(sd.keys()) == list(keys.keys())
    path = str(tmp_path / "m.pth")
    torch.save({"epoch": 1, "state_dict": {k: v.detach().cpu().contiguous().clone() for k, v in sd.items()}}, path)
    enc2 = _seeded_encoder()
    with torch.no_grad():
        for p in enc2.parameters():
            p.zero_()
    enc2.load_state_dict(torch.load(path)["state_dict"])
    for (k, a), (_, b) in zip(enc.state_dict().items(), enc2.state_dict().items()):
        assert torch.equal(a.cpu(), b.cpu()), k
        assert list(a.shape) == keys[k]


def test_moco_three_steps_match_reference(golden):
    """MoCo.forward + CE + SGD for 3 steps at the reference run's lr 0.05.  Step 0 is compared with the reference run
    (moco_3steps.npz) tightly.  Later steps amplify fp32 noise chaotically (batch-8 BatchNorms behind activations of 1e7 ..
    1e10: the CPU oracle itself drifts 4e-2 from the reference's logits by step 2), so every step is ALSO compared with the
    oracle restarted from the GPU's own state and arbitrated by the same oracle in float64.
    Round 4: no skipped step and no fixed allowance any more.  (a) The oracles take the GPU's ReLU decisions
    (hipops.RELU_TAP -> conftest.gpu_relu_decisions): where a unit sits within the forward pass's own fp32 error of zero,
    "fires or not" is the evaluator's to decide, and one unit of the head moves every upstream gradient by percents (round 3
    skipped the gradient comparison of such steps); every unit where the GPU differs from float64 is checked to be on that
    edge.  (b) The size of the fp32 error on a later step is MEASURED - the worst of four CPU fp32 evaluations of the same
    step, inputs 0 / +1 / -1 / +2 roundings away - where round 3 allowed a flat 0.15."""
    import copy
    from conftest import f32_equivalent, gpu_relu_decisions, assert_relu_flips_on_edge
    from oracle import train_ref as T
    from test_oracle_train import seeded_sd
    from cet_pick_amd.models.moco import MoCo
    from cet_pick_amd import hipops as H
    g = golden("moco_3steps.npz")
    torch.manual_seed(7)
    q, k = _seeded_encoder(), _seeded_encoder()
    moco = MoCo(q, k, dim=128, r=64, m=0.99, T=0.1).cuda()
    assert list(moco.state_dict().keys()) == [str(s) for s in g["state_keys"]]
    moco.queue.copy_(torch.from_numpy(g["queue0"]).cuda())
    aq, ak = moco.flatten_parameters()
    ref = T.MocoRef(seeded_sd(), torch.from_numpy(g["queue0"]), m=0.99, T=0.1, lr=0.05)
    ref64 = T.MocoRef({k: (v.double() if v.is_floating_point() else v) for k, v in seeded_sd().items()},
                      torch.from_numpy(g["queue0"]).double(), m=0.99, T=0.1, lr=0.05)
    gen = torch.Generator().manual_seed(123)
    torch.randn(128, 64, generator=gen)
    B = 8
    moco.train()
    all_flips = []
    for step in range(3):
        im_q = torch.randn(B, 1, 32, 32, 32, generator=gen)
        im_k = im_q.flip(4) + 0.1 * torch.randn(B, 1, 32, 32, 32, generator=gen)
        masks = gpu_relu_decisions(moco.encoder_q, im_q.cuda())      # the decisions of the forward pass below
        aq.zero_grad()
        logits, labels = moco(im_q.cuda(), im_k.cuda())
        loss = H.cross_entropy_label0(logits)
        loss.backward()
        lg = logits.detach().cpu()
        # the spread of fp32 itself on this step: the same oracle step from the same state, inputs one and two roundings away
        spread = [copy.deepcopy(ref).step(im_q * (1.0 + eps), im_k * (1.0 + eps), relu_masks=masks)[2]
                  for eps in (2.0 ** -23, -2.0 ** -23, 2.0 ** -22)]
        # the natural branches (no masks): where do the GPU's decisions differ, and is that on the edge?
        pre32, pre64 = {}, {}
        copy.deepcopy(ref).step(im_q, im_k, pre=pre32)
        copy.deepcopy(ref64).step(im_q.double(), im_k.double(), pre=pre64)
        flips = assert_relu_flips_on_edge(masks, pre64, pre32)
        all_flips.append({k_: v for k_, v in flips.items() if v})
        assert step > 0 or not all_flips[0].get("proj.1") and not all_flips[0].get("proj.4")
        l_ref, loss_ref, grads = ref.step(im_q, im_k, relu_masks=masks)
        l64, loss64, grads64 = ref64.step(im_q.double(), im_k.double(), relu_masks=masks)
        # the outputs are cosines: logits * T.  1e-3 on them at every step (north_star), against float64
        np.testing.assert_allclose(0.1 * lg.numpy(), 0.1 * l64.numpy(), rtol=0, atol=1e-3)
        f32_equivalent(lg.numpy(), l_ref.numpy(), l64.numpy(), what="logits step %d" % step)
        # (a scalar: the ratio of two fp32 evaluations' errors is unbounded when the CPU's happens to be ~0, so behind the
        # arbiter stands north_star's 1e-3 - in the exploding regime of steps 1-2 the logits above are the sharper check)
        assert abs(float(loss.detach()) - loss64) <= max(2 * abs(loss_ref - loss64) + 2e-5, 1e-3 if step > 0 else 0.0)
        gscale = float(sum(float(v.norm()) ** 2 for v in grads64.values()) ** 0.5)   # whole-gradient norm
        for n, p in moco.encoder_q.named_parameters():
            if n == "fc.bias" or n not in grads:
                continue
            a = p.grad.detach().cpu().contiguous()
            # (floor: parameters whose gradient is noise next to the rest, e.g. the bias in front of a BatchNorm)
            floor = 5e-5 * gscale / (float(grads64[n].norm()) + 1e-30) + 2e-6
            f32_equivalent(a.numpy(), grads[n].numpy(), grads64[n].numpy(), floor=floor, what="step %d grad %s" % (step, n),
                           more_cpu32=[sp[n].numpy() for sp in spread])
            if step == 0:          # well-conditioned seeded weights: also tight in absolute terms
                assert float((a - grads[n]).norm()) <= 2e-4 * float(grads[n].norm()) + 5e-5 * gscale + 1e-6, n
        if step == 0:     # later steps of the reference run are pinned through the oracle (CPU test)
            np.testing.assert_allclose(lg.numpy(), g["logits_0"], rtol=0, atol=1e-3)
            assert abs(float(loss.detach()) - float(g["loss_0"])) < 1e-3
        assert int(moco.queue_ptr) == int(g[f"ptr_{step}"])
        assert labels.dtype == torch.long and int(labels.sum()) == 0
        w_before = aq.flat.clone()
        H.sgd_step_(aq.flat, aq.flat_grad, 0.05)
        np.testing.assert_allclose(moco.queue.cpu().numpy(), ref.queue.numpy(), rtol=0, atol=1e-4)
        # SGD applied exactly the gradient that was just compared (the gradient itself is held to the float64 arbiter above:
        # in the exploding regime of steps 1 - 2 a fixed tolerance on the updated weights against the ORACLE's weights would
        # only restate the CPU's own fp32 error, 3e-2 of a gradient norm of ~20)
        want = w_before - 0.05 * aq.flat_grad
        assert float((aq.flat - want).abs().max()) <= 2e-7 * float(w_before.abs().max()) + 1e-9, step
        for n, p in moco.encoder_k.named_parameters():
            np.testing.assert_allclose(p.detach().cpu().contiguous().numpy(), ref.k[n].numpy(), rtol=0, atol=1e-4, err_msg=n)
        # restart both oracles from the GPU's state
        for enc, dst, dst64 in ((moco.encoder_q, ref.q, ref64.q), (moco.encoder_k, ref.k, ref64.k)):
            for n, t in list(enc.named_parameters()) + list(enc.named_buffers()):
                dst[n] = t.detach().cpu().contiguous().clone()
                dst64[n] = dst[n].double() if dst[n].is_floating_point() else dst[n].clone()
        ref.queue = moco.queue.cpu().clone()
        ref64.queue = ref.queue.double()
        ref64.ptr = ref.ptr
    np.testing.assert_allclose(moco.encoder_k.fc.weight.detach().cpu().numpy(), g["k_fc_weight"], rtol=0, atol=1e-3)
    print("units decided differently from float64 (all verified to be on the edge), per step:", all_flips)


def test_moco_three_wellconditioned_steps_match_reference(golden):
    """The reference's own three MoCo steps at a well-conditioned learning rate (moco_3steps_wc.npz;
    gen_golden.py::gen_moco_wc): logits, loss, pointer, the norm of EVERY parameter gradient and samples of nine of them are
    compared on EVERY step, directly with the reference - no restart, no skipped step.
    Why lr 1e-5 and not the bench's 1e-3: with these weights the step map amplifies any perturbation ~100x per step at 1e-3
    (CPU fp32 against float64 of the same oracle: 2e-5, 5e-3, 3e-1 relative on the stem gradients at steps 0, 1, 2, batch 8
    and batch 32 alike; measured on the GPU against the reference: 3e-5, 5e-3, 3e-1 - tools/diag_wc_steps.py), so from step 1
    on no two fp32 evaluations agree to 1e-3 there; at 1e-5 they stay 2e-5 apart and every step is held to 1e-3.
    Sampled gradient entries (round 4): a unit within rounding of its ReLU edge fires differently in two valid fp32
    evaluations and moves single upstream entries by 1e-3 .. 2e-3 (round 3 allowed 3e-3 of the largest sample for that).  Now
    the float64 oracle runs next to the GPU and is evaluated on BOTH branches - its own, which is the reference's, and the
    GPU's (its tapped ReLU decisions, each differing unit verified to be on the edge): the GPU's samples are held to 2e-4
    against float64 on the GPU's branch, and to the reference's samples within 1e-3 + what float64 says the branches differ by."""
    import copy
    from conftest import gpu_relu_decisions, assert_relu_flips_on_edge
    from oracle import train_ref as T
    from test_oracle_train import seeded_sd
    from cet_pick_amd.models.moco import MoCo
    from cet_pick_amd import hipops as H
    g = golden("moco_3steps_wc.npz")
    lr = float(g["lr"])
    torch.manual_seed(7)
    moco = MoCo(_seeded_encoder(), _seeded_encoder(), dim=128, r=64, m=0.99, T=0.1).cuda()
    moco.queue.copy_(torch.from_numpy(g["queue0"]).cuda())
    aq, ak = moco.flatten_parameters()
    q0, k0 = aq.flat.double().clone(), ak.flat.double().clone()
    gen = torch.Generator().manual_seed(123)
    torch.randn(128, 64, generator=gen)
    idx = g["sample_idx"]
    moco.train()
    compared = 0
    ref64 = T.MocoRef({k_: (v.double() if v.is_floating_point() else v) for k_, v in seeded_sd().items()},
                      torch.from_numpy(g["queue0"]).double(), m=0.99, T=0.1, lr=lr)
    for step in range(3):
        im_q = torch.randn(8, 1, 32, 32, 32, generator=gen)
        im_k = im_q.flip(4) + 0.1 * torch.randn(8, 1, 32, 32, 32, generator=gen)
        masks = gpu_relu_decisions(moco.encoder_q, im_q.cuda())
        _, _, g64b = copy.deepcopy(ref64).step(im_q.double(), im_k.double(), relu_masks=masks)     # float64, the GPU's branch
        pre64 = {}
        _, _, g64 = ref64.step(im_q.double(), im_k.double(), pre=pre64)                            # float64's own = the run's state
        assert_relu_flips_on_edge(masks, pre64)
        aq.zero_grad()
        logits, labels = moco(im_q.cuda(), im_k.cuda())
        loss = H.cross_entropy_label0(logits)
        loss.backward()
        np.testing.assert_allclose(logits.detach().cpu().numpy(), g[f"logits_{step}"], rtol=0, atol=1e-3, err_msg="step %d" % step)
        assert abs(float(loss.detach()) - float(g[f"loss_{step}"])) < 1e-4
        assert int(moco.queue_ptr) == int(g[f"ptr_{step}"])
        for n, p in moco.encoder_q.named_parameters():
            if f"gnorm_{step}_{n}" not in g.files:
                continue
            gf = p.grad.detach().cpu().contiguous().reshape(-1).numpy()
            want = float(g[f"gnorm_{step}_{n}"])
            if want > 1e-4:          # (fc.bias sits in front of a batch-statistics BatchNorm: its gradient is rounding noise)
                assert abs(np.linalg.norm(gf.astype(np.float64)) - want) <= 1e-3 * want, (step, n)
                compared += 1
                # the whole gradient against float64 on the GPU's branch
                r = g64b[n].reshape(-1).numpy()
                assert np.linalg.norm(gf - r) <= 2e-4 * np.linalg.norm(r), (step, n, np.linalg.norm(gf - r) / np.linalg.norm(r))
            if f"gsample_{step}_{n}" in g.files:
                ws = g[f"gsample_{step}_{n}"]
                sel = idx % gf.size
                branch = np.abs(g64b[n].reshape(-1).numpy()[sel] - g64[n].reshape(-1).numpy()[sel])
                np.testing.assert_array_less(np.abs(gf[sel] - ws), 1e-3 * float(np.abs(ws).max()) + branch + 1e-7,
                                             err_msg="step %d %s" % (step, n))
        H.sgd_step_(aq.flat, aq.flat_grad, lr)
    assert compared >= 3 * 28
    # SGD and EMA ran: the weight DELTAS over the three steps (an update is ~1e-6 of a weight, so the weights themselves
    # would pass with no update at all; the deltas are a few ulps of the weights, hence 2e-2 of the largest)
    qd, kd = aq.flat.double() - q0, ak.flat.double() - k0
    for (n, p), off in zip(moco.encoder_q.named_parameters(), aq.offsets):
        for which, dflat in (("q", qd), ("k", kd)):
            key = f"{which}_delta_{n}"
            if key not in g.files:
                continue
            stride = {"fc.weight": 7, "layer1.0.conv1.weight": 997, "layer3.0.downsample.0.weight": 101}[n]
            view = torch.as_strided(dflat, p.shape, p.stride(), off).cpu().contiguous().reshape(-1)[::stride].numpy()
            want = g[key]
            assert np.abs(want).max() > 0
            if which == "k":
                # k moves by (1 - m) of q's update per step: single ulps of a weight - the EMA is held to one ulp instead
                np.testing.assert_allclose(view, want, rtol=0, atol=1.2e-7, err_msg=key)
                continue
            np.testing.assert_allclose(view, want, rtol=0, atol=2e-2 * float(np.abs(want).max()), err_msg=key)
    np.testing.assert_allclose(moco.queue.cpu().numpy(), g["queue_final"], rtol=0, atol=1e-4)


@pytest.mark.parametrize("m,ci,co,bias", [(64, 256, 128, True), (64, 128, 128, False), (5, 48, 32, True), (2048, 256, 128, True),
                                          (3, 48, 16, False), (70, 16, 16, True), (1, 1008, 64, True)])
def test_linear_fwd_bwd(m, ci, co, bias):
    """nn.Linear (fc / projection head): one register-staged launch each for forward (bias in the epilogue), data gradient
    and weight gradient (conv_cube2.hip small_gemm_kernel) - ragged sizes exercise its row / column / k masks; 2048 rows
    take the implicit GEMM with the bias in its epilogue."""
    from cet_pick_amd import hipops as H
    g = torch.Generator().manual_seed(m + ci + co)
    lin = H.HipLinear(ci, co, bias=bias).cuda()
    ref = torch.nn.Linear(ci, co, bias=bias)
    with torch.no_grad():
        ref.weight.copy_(torch.randn(co, ci, generator=g) * 0.1)
        lin.weight.copy_(ref.weight.cuda())
        if bias:
            ref.bias.copy_(torch.randn(co, generator=g))
            lin.bias.copy_(ref.bias.cuda())
    x = torch.randn(m, ci, generator=g)
    dy = torch.randn(m, co, generator=g)
    xg = x.cuda().requires_grad_(True)
    xr = x.clone().requires_grad_(True)
    y = lin(xg)
    yr = ref(xr)
    np.testing.assert_allclose(y.detach().cpu().numpy(), yr.detach().numpy(), rtol=1e-4, atol=1e-4)
    y.backward(dy.cuda())
    yr.backward(dy)
    np.testing.assert_allclose(xg.grad.cpu().numpy(), xr.grad.numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(lin.weight.grad.cpu().numpy(), ref.weight.grad.numpy(), rtol=1e-4, atol=1e-3)
    if bias:
        np.testing.assert_allclose(lin.bias.grad.cpu().numpy(), ref.bias.grad.numpy(), rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("shape", [(64, 2, 2, 2, 256), (5, 4, 4, 4, 32), (7, 1, 1, 1, 64), (3, 3, 2, 2, 16)])
def test_fused_bn_relu_avgpool_equals_separate_kernels(shape):
    """feature_3d's BatchNorm + ReLU + global average pool in one launch each way: bit-identical to the separate
    kernels (same summation orders); a voxel count that is not a power of two takes the separate kernels."""
    from cet_pick_amd import hipops as H
    g = torch.Generator().manual_seed(sum(shape))
    c = shape[-1]
    x0 = (torch.randn(shape, generator=g) * 2 + 0.3).cuda()
    dp = torch.randn(shape[0], c, generator=g).cuda()
    outs = []
    for fused in (True, False):
        bn = H.HipBatchNorm(c).cuda()
        with torch.no_grad():
            bn.weight.copy_(torch.linspace(0.5, 1.5, c)); bn.bias.copy_(torch.linspace(-0.2, 0.2, c))
        x = x0.clone().requires_grad_(True)
        p = H.bn_relu_global_avgpool(x, bn) if fused else H.global_avgpool(bn(x, relu=True))
        p.backward(dp)
        outs.append((p.detach(), x.grad, bn.weight.grad, bn.bias.grad, bn.running_mean.clone(), bn.running_var.clone()))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    ref = torch.nn.functional.relu(torch.nn.functional.batch_norm(
        x0.cpu().permute(0, 4, 1, 2, 3), None, None, torch.linspace(0.5, 1.5, c), torch.linspace(-0.2, 0.2, c), True)).mean((2, 3, 4))
    np.testing.assert_allclose(outs[0][0].cpu().numpy(), ref.numpy(), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("shape,train", [((3, 10, 12, 14, 16), True), ((2, 9, 7, 8, 64), True), ((2, 8, 8, 8, 16), False)])
def test_fused_bn_relu_maxpool_matches_unfused(shape, train):
    """stem fusion: maxpool3d(relu(bn(x))) forward / backward against torch (BatchNorm3d + ReLU + MaxPool3d)."""
    import numpy as np
    from cet_pick_amd import hipops as H
    n, d, h, w, c = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(n, c, d, h, w, generator=g) * 2 + 0.5
    bn_ref = torch.nn.BatchNorm3d(c)
    with torch.no_grad():
        bn_ref.weight.copy_(torch.rand(c, generator=g) + 0.5)
        bn_ref.bias.copy_(torch.randn(c, generator=g) * 0.3)
        bn_ref.running_mean.copy_(torch.randn(c, generator=g) * 0.1)
        bn_ref.running_var.copy_(torch.rand(c, generator=g) + 0.5)
    bn = H.HipBatchNorm(c)
    bn.load_state_dict(bn_ref.state_dict())
    bn = bn.cuda()
    bn_ref.train(train); bn.train(train)
    xr = x.clone().requires_grad_(True)
    yr = torch.nn.functional.max_pool3d(torch.relu(bn_ref(xr)), 3, 2, 1)
    xc = x.permute(0, 2, 3, 4, 1).contiguous().cuda().requires_grad_(train)
    y = H.bn_relu_maxpool3d(xc, bn, 3, 2, 1)
    np.testing.assert_allclose(y.detach().permute(0, 4, 1, 2, 3).cpu().numpy(), yr.detach().numpy(), rtol=1e-5, atol=1e-5)
    if train:
        dy = torch.randn(yr.shape, generator=g)
        yr.backward(dy)
        y.backward(dy.permute(0, 2, 3, 4, 1).contiguous().cuda())
        np.testing.assert_allclose(xc.grad.permute(0, 4, 1, 2, 3).cpu().numpy(), xr.grad.numpy(), rtol=1e-4, atol=2e-5)
        np.testing.assert_allclose(bn.weight.grad.cpu().numpy(), bn_ref.weight.grad.numpy(), rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(bn.bias.grad.cpu().numpy(), bn_ref.bias.grad.numpy(), rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(bn.running_var.cpu().numpy(), bn_ref.running_var.numpy(), rtol=1e-5, atol=1e-6)
        assert int(bn.num_batches_tracked) == 1


@pytest.mark.gpu
@pytest.mark.parametrize("n,size", [(3, 32), (2, 48), (64, 32)])
def test_stem_epilogue_statistics_match_statistics_pass(n, size):
    """mi_conv3d_stem_stats_f32: the output equals the plain stem convolution's bit for bit and the BatchNorm sums from
    its epilogue equal float64 column sums of that output (to float32 rounding of the 16-voxel partial sums); the encoder's
    trunk produces the same activations with and without them (MI_STEM_NO_STATS=1)."""
    import ctypes
    import numpy as np
    from cet_pick_amd import hipops as H, _lib as L
    lib = L.lib()
    g = torch.Generator().manual_seed(n * 100 + size)
    x = (torch.randn(n, size, size, size, 1, generator=g) * 1.5 + 0.2).cuda()
    conv = H.HipConv3d(1, 64, 7, 2, 3).cuda()
    y_plain = conv(x)
    nbytes = lib.mi_conv3d_stem_stats_workspace_bytes(n, size, size, size, 64)
    assert nbytes > 0
    ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    y = torch.empty_like(y_plain)
    sums = torch.zeros(128, dtype=torch.float64, device="cuda")
    rc = lib.mi_conv3d_stem_stats_f32(L.ptr(x), L.ptr(conv.weight), L.ptr(y), n, size, size, size, 64, L.ptr(sums), L.ptr(ws),
                                      ws.numel(), L.stream())
    assert rc == 0
    assert torch.equal(y, y_plain.detach())
    y64 = y.double().reshape(-1, 64)
    want = torch.cat([y64.sum(0), (y64 * y64).sum(0)]).cpu().numpy()
    got = sums.cpu().numpy()
    scale = np.concatenate([y64.abs().sum(0).cpu().numpy(), want[64:]])
    assert np.max(np.abs(got - want) / scale) < 2e-6          # fp32 partial sums over 16 voxels, doubles from there
    # an unsupported geometry is declined, not mis-run
    assert lib.mi_conv3d_stem_stats_workspace_bytes(n, size + 2, size, size, 64) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("m,ci,co,relu,bias", [(64, 128, 128, True, False), (16, 128, 256, False, False), (37, 256, 128, True, True)])
def test_linear_bn_one_launch_matches_two_launches_and_torch(m, ci, co, relu, bias, monkeypatch):
    """hipops.linear_bn (mi_linear_bn_fwd_f32: Linear + BatchNorm1d (+ ReLU) in the product's epilogue) against the two
    separate launches (MI_NO_LINEAR_BN=1) and against torch in float64: outputs, running statistics, every gradient."""
    import numpy as np
    from cet_pick_amd import hipops as H
    g = torch.Generator().manual_seed(m + ci + co)
    x0 = torch.randn(m, ci, generator=g)
    dy = torch.randn(m, co, generator=g)
    w0 = torch.randn(co, ci, generator=g) * 0.1
    b0 = torch.randn(co, generator=g) * 0.1
    gam, bet = torch.rand(co, generator=g) + 0.5, torch.randn(co, generator=g) * 0.2

    def run(fused):
        monkeypatch.setenv("MI_NO_LINEAR_BN", "0" if fused else "1")
        lin = H.HipLinear(ci, co, bias=bias).cuda()
        bn = H.HipBatchNorm(co).cuda()
        with torch.no_grad():
            lin.weight.copy_(w0.cuda())
            if bias:
                lin.bias.copy_(b0.cuda())
            bn.weight.copy_(gam.cuda()); bn.bias.copy_(bet.cuda())
        x = x0.cuda().requires_grad_(True)
        y = H.linear_bn(x, lin, bn, relu=relu)
        y.backward(dy.cuda())
        out = [y.detach(), x.grad, lin.weight.grad, bn.weight.grad, bn.bias.grad, bn.running_mean, bn.running_var]
        if bias:
            out.append(lin.bias.grad)
        assert int(bn.num_batches_tracked) == 1
        return [t.detach().cpu().double().numpy() for t in out]

    one, two = run(True), run(False)
    # float64 reference
    lin_r = torch.nn.Linear(ci, co, bias=bias).double()
    bn_r = torch.nn.BatchNorm1d(co).double()
    with torch.no_grad():
        lin_r.weight.copy_(w0.double())
        if bias:
            lin_r.bias.copy_(b0.double())
        bn_r.weight.copy_(gam.double()); bn_r.bias.copy_(bet.double())
    xr = x0.double().requires_grad_(True)
    yr = bn_r(lin_r(xr))
    yr = torch.relu(yr) if relu else yr
    yr.backward(dy.double())
    ref = [yr.detach(), xr.grad, lin_r.weight.grad, bn_r.weight.grad, bn_r.bias.grad, bn_r.running_mean, bn_r.running_var]
    if bias:
        ref.append(lin_r.bias.grad)
    ref = [t.detach().numpy() for t in ref]
    for a, b, r in zip(one, two, ref):
        scale = max(np.abs(r).max(), 0.05)                   # (a Linear bias in front of a BatchNorm has a zero gradient)
        assert np.abs(a - b).max() <= 2e-6 * scale           # same arithmetic, another summation order of the statistics
        assert np.abs(a - r).max() <= 2e-5 * scale


@pytest.mark.gpu
@pytest.mark.parametrize("n,c", [(64, 256), (5, 256), (70, 128)])
def test_cube2_final_launch_equals_partial_sums_plus_reduce(n, c, monkeypatch):
    """mi_conv3d_cube2_f32 (layer3 / feature_3d: the last of a tile's four reduction-quarter workgroups sums them in slab order and
    applies the epilogue) against the same kernel with partial sums + reduce launch (MI_CUBE2_REDUCE=1): forward with residual +
    ReLU and data gradient with residual + mask BIT FOR BIT - the sum must not depend on who arrives last -, on two streams at
    once (one workspace per stream), and the arrival counters are zero again after every call (moco_encoder_3d.py:55-84,172,178)."""
    from cet_pick_amd import hipops as H, _lib as L
    g = torch.Generator().manual_seed(n + c)
    x = cl(torch.randn(n, c, 2, 2, 2, generator=g))
    res = cl(torch.randn(n, c, 2, 2, 2, generator=g))
    mask = cl(torch.randn(n, c, 2, 2, 2, generator=g))
    dy = cl(torch.randn(n, c, 2, 2, 2, generator=g))
    param, _ = make_w(c, c, 3, g)
    monkeypatch.setenv("MI_CUBE2_REDUCE", "1")
    yf0 = H.conv_fwd(x, param, 3, 1, 1, res, True)
    yd0 = H.conv_dgrad(dy, param, (n, 2, 2, 2, c), 3, 1, 1, res, mask)
    assert L.lib().mi_debug_last_conv_kernel().decode() == "cube2 + reduce"
    monkeypatch.delenv("MI_CUBE2_REDUCE")
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    outs = []
    for rep in range(3):
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            yd1 = H.conv_dgrad(dy, param, (n, 2, 2, 2, c), 3, 1, 1, res, mask)
        yf1 = H.conv_fwd(x, param, 3, 1, 1, res, True)
        outs.append((yf1, yd1))
    assert L.lib().mi_debug_last_conv_kernel().decode() == "cube2"
    torch.cuda.synchronize()
    for yf1, yd1 in outs:
        assert torch.equal(yf1, yf0) and torch.equal(yd1, yd0)
    tickets = [v for k, v in L._workspaces.items() if k[1] == "cube2"]
    assert len(tickets) >= 2                                  # one workspace per stream
    for ws in tickets:
        assert int(ws[:16384].view(torch.int32).abs().sum()) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(4, 16, 16, 16, 64), (2, 5, 7, 16, 64), (3, 10, 12, 32, 32), (2, 2, 2, 64, 16), (1, 9, 3, 8, 128)])
def test_maxpool_backward_parity_form_equals_generic(shape, monkeypatch):
    """maxpool_bwd_k3s2_kernel (the stem's pool: a workgroup per input plane, pooled planes in LDS, candidates fixed by coordinate
    parity; planes of Wi * C / 4 = 256 vectors) against the generic gather kernel (MI_MAXPOOL_BWD_GENERIC=1): BIT-identical - same
    candidates, same order of additions - and both against torch's max_pool3d backward, also on odd depths / heights
    (moco_encoder_3d.py:169)."""
    from cet_pick_amd import hipops as H
    n, d, h, w, c = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(n, c, d, h, w, generator=g)
    xg = cl(x).requires_grad_(True)
    y = H.maxpool3d(xg, 3, 2, 1)
    dy = torch.randn(y.shape, generator=g).cuda()
    (g_par,) = torch.autograd.grad(y, xg, dy, retain_graph=True)
    monkeypatch.setenv("MI_MAXPOOL_BWD_GENERIC", "1")
    (g_gen,) = torch.autograd.grad(y, xg, dy)
    assert torch.equal(g_par, g_gen)
    xr = x.clone().requires_grad_(True)
    yr = F.max_pool3d(xr, 3, 2, 1)
    yr.backward(dy.cpu().permute(0, 4, 1, 2, 3))
    np.testing.assert_allclose(ncdhw(g_par).numpy(), xr.grad.numpy(), rtol=0, atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(4, 16, 16, 16, 64), (2, 5, 7, 16, 64), (3, 10, 12, 32, 32), (2, 2, 2, 64, 16), (1, 9, 3, 8, 128)])
def test_stem_backward_matches_torch(shape):
    """The backward of maxpool3d(relu(bn(x)), 3, 2, 1) - pool backward, column reduce with the ReLU mask recomputed from x, apply -
    against torch (BatchNorm3d + ReLU + MaxPool3d), on the stem's geometry class incl. odd depths / heights and a last band of fewer
    than four rows (moco_encoder_3d.py:170-172)."""
    from cet_pick_amd import hipops as H
    n, d, h, w, c = shape
    g = torch.Generator().manual_seed(sum(shape) + 5)
    x = torch.randn(n, c, d, h, w, generator=g) * 2 + 0.5
    bn_ref = torch.nn.BatchNorm3d(c)
    with torch.no_grad():
        bn_ref.weight.copy_(torch.rand(c, generator=g) + 0.5)
        bn_ref.bias.copy_(torch.randn(c, generator=g) * 0.3)
    xr = x.clone().requires_grad_(True)
    yr = F.max_pool3d(torch.relu(bn_ref(xr)), 3, 2, 1)
    dy = torch.randn(yr.shape, generator=g)
    yr.backward(dy)
    bn = H.HipBatchNorm(c)
    bn.load_state_dict(bn_ref.state_dict())
    bn = bn.cuda().train()
    xc = cl(x).requires_grad_(True)
    y = H.bn_relu_maxpool3d(xc, bn, 3, 2, 1)
    y.backward(cl(dy))
    np.testing.assert_allclose(ncdhw(xc.grad).numpy(), xr.grad.numpy(), rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(bn.weight.grad.cpu().numpy(), bn_ref.weight.grad.numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(bn.bias.grad.cpu().numpy(), bn_ref.bias.grad.numpy(), rtol=1e-4, atol=1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("case", [(64, 4, 128, 128, 3, 1), (70, 4, 128, 128, 3, 1), (9, 8, 64, 128, 3, 2), (9, 8, 64, 128, 1, 2),
                                  (5, 6, 64, 64, 3, 2), (33, 2, 256, 128, 3, 1), (3, 8, 64, 64, 3, 1)])
def test_pair_weight_gradient_forms_match_float64(case, monkeypatch):
    """pair_wgrad_kernel with a tap's chain in segments (MI_PAIR_WGRAD_MAXOUT) on layer2 / layer2.0 / layer3 / layer1 shapes against
    float64, with the error bound of the default path (bf16x3 = f32-equivalent)."""
    from cet_pick_amd import hipops as H, _lib as L
    n, d, ci, co, k, s = case
    pad = 1 if k == 3 else 0
    monkeypatch.setenv("MI_PAIR_WGRAD_MAXOUT", "8")
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(n, ci, d, d, d, generator=g)
    param, w = make_w(co, ci, k, g)
    x64 = x.double().cuda()
    w64 = w.double().cuda().requires_grad_(True)
    y64 = F.conv3d(x64, w64, stride=s, padding=pad)
    dy = torch.randn(y64.shape, generator=g)
    (gw,) = torch.autograd.grad(y64, w64, dy.double().cuda())
    param.grad = None
    H.conv_wgrad_into(cl(x), cl(dy), param, k, s, pad)
    assert L.lib().mi_debug_last_conv_kernel().decode().startswith("pair_wgrad")
    scale = float(gw.abs().max())
    assert float((param.grad.double() - gw).abs().max()) <= 2e-6 * scale * max(1.0, (n * d ** 3 / 512) ** 0.5)


@pytest.mark.gpu
@pytest.mark.parametrize("case,nb,family", [((16, 8, 64, 64, 3, 1), 4, "direct3_wgrad x nb"), ((16, 8, 64, 64, 3, 1), 3, "direct3_wgrad x nb"),
                                            ((24, 4, 128, 128, 3, 1), 3, "direct3s_wgrad x nb"), ((64, 4, 128, 128, 3, 1), 2, "direct3s_wgrad x nb"), ((7, 4, 128, 128, 3, 1), 4, "direct3s_wgrad x nb"),
                                            ((24, 4, 128, 64, 3, 1), 3, "implicit GEMM x nb"), ((24, 4, 128, 128, 3, 1), 3, "implicit GEMM x nb"), ((20, 2, 256, 256, 3, 1), 4, "pair_wgrad x nb"),
                                            ((6, 8, 32, 48, 3, 1), 2, "implicit GEMM x nb"), ((16, 8, 64, 64, 3, 1), 5, "direct3_wgrad x nb"),
                                            #  the 64^3-crop shapes: no batched kernel - the group goes out as single launches of the dedicated ones
                                            ((6, 16, 64, 64, 3, 1), 2, "single:direct3_wgrad (8 x 8 tiles)"), ((16, 8, 128, 128, 3, 1), 3, "single:direct3_wgrad (128 channels)")])
def test_batched_weight_gradients_against_single_launches(case, nb, family, monkeypatch):
    """Round 5: the weight gradients of a stage's equal convolutions in ONE launch (hipops.run_wgrad_jobs ->
    mi_convnd_wgrad_slabs_batch_f32; layer1's four on direct3_wgrad_kernel, layer2's three on the implicit GEMM, layer3's on
    pair_wgrad_kernel) against one launch per convolution.  Implicit GEMM and pair_wgrad: same chains, same slabs, same reduce -
    EQUAL BIT FOR BIT.  direct3_wgrad cuts a batched problem into 7 chains per (dz, dy) pair instead of 28 (that is where the time
    goes): equal to float64 within the single launch's bound, and a problem's gradient does not depend on its neighbours in the
    launch (the same convolution in a launch of two, bit for bit).  Five jobs = a launch of four and a single one.
    (moco_encoder_3d.py:55-84)"""
    from cet_pick_amd import hipops as H, _lib as L
    n, d, ci, co, k, s_ = case
    pad = 1
    if family.startswith("direct3s"):
        monkeypatch.setenv("MI_D3S_WGRAD", "1")        # layer2's shape on direct3_wgrad_kernel<true>: opt-in (the default: implicit GEMM)
    g = torch.Generator().manual_seed(sum(case) + nb)
    xs = [cl(torch.randn(n, ci, d, d, d, generator=g)) for _ in range(nb)]
    do = (d + 2 * pad - k) // s_ + 1
    dys = [cl(torch.randn(n, co, do, do, do, generator=g)) for _ in range(nb)]
    single, batch, pair = [[make_w(co, ci, k, g)[0] for _ in range(nb)] for _ in range(3)]

    def batched(params, idx):
        H.SIDE_WGRADS = []
        for i in idx:
            params[i].grad = None
            H.conv_wgrad_into(xs[i], dys[i], params[i], k, s_, pad)
        assert len(H.SIDE_WGRADS) == len(idx) and all(isinstance(j, H._WgradJob) for j in H.SIDE_WGRADS)
        H.run_wgrad_jobs(H.SIDE_WGRADS)
        H.SIDE_WGRADS = None
        name = L.lib().mi_debug_last_conv_kernel().decode()
        H.flush_wgrad_reduces()
        return name

    H.DEFERRED_WGRADS = []
    try:
        for x, dy, prm in zip(xs, dys, single):
            prm.grad = None
            H.conv_wgrad_into(x, dy, prm, k, s_, pad)
        H.flush_wgrad_reduces()
        name = batched(batch, list(range(nb)))
        assert name.startswith(family[7:] if family.startswith("single:") else family if nb != 5 else "direct3_wgrad + reduce"), name
        batched(pair, [0, nb - 1])                     # the first and the last problem again, in a launch of two
    finally:
        H.DEFERRED_WGRADS = None
        H.SIDE_WGRADS = None
    torch.cuda.synchronize()
    for i, (a, b) in enumerate(zip(single, batch)):
        x64 = ncdhw(xs[i]).double().cuda()
        w64 = torch.zeros(co, ci, k, k, k, dtype=torch.float64, device="cuda", requires_grad=True)
        (gw,) = torch.autograd.grad(F.conv3d(x64, w64, stride=s_, padding=pad), w64, ncdhw(dys[i]).double().cuda())
        bound = 2e-6 * float(gw.abs().max()) * max(1.0, (n * do ** 3 / 512) ** 0.5)
        assert float((b.grad.double() - gw).abs().max()) <= bound
        assert float((a.grad.double() - gw).abs().max()) <= bound
        if not family.startswith("direct3") or (nb == 5 and i == 4):         # (single launches either way: bit for bit)
            assert torch.equal(a.grad, b.grad), i
        if i in (0, nb - 1) and nb != 5:
            assert torch.equal(pair[i].grad, b.grad), i


@pytest.mark.gpu
def test_conv_dispatch_by_shape(monkeypatch):
    """Which kernel family a convolution call takes (mi_debug_last_conv_kernel; tools/bench_conv.py --kernels prints the table):
    the encoder's shapes at the benchmark's crop size take the patch-resident kernels, other shapes the implicit GEMM, and the
    A/B switches move them."""
    from cet_pick_amd import hipops as H, _lib as L
    lib = L.lib()
    def fwd(n, d, ci, co, k=3, s=1, p=1):
        x = torch.randn(n, d, d, d, ci, device="cuda")
        w = H.conv_weight_param(co, ci, k); w.data = w.data.cuda(); w.data.normal_()
        H.conv_fwd(x, w, k, s, p)
        return lib.mi_debug_last_conv_kernel().decode()
    monkeypatch.delenv("MI_CONV_NO_DIRECT", raising=False)
    assert fwd(4, 32, 1, 64, 7, 2, 3) == "stem_fwd"
    assert fwd(4, 8, 64, 64) == "direct3"
    assert fwd(4, 4, 128, 128).startswith("direct3s")
    assert fwd(4, 2, 256, 256).startswith("cube2")
    assert fwd(2, 16, 64, 64).startswith("implicit GEMM")          # layer1 of a 64^3 crop
    assert fwd(16, 8, 128, 128) == "direct3 (128 channels)"        # layer2 of a 64^3 crop (round 4), from 128 workgroups on
    assert fwd(2, 8, 128, 128).startswith("implicit GEMM")
    assert fwd(16, 4, 256, 256) == "direct3s (256 channels)"       # layer3 of a 64^3 crop (round 4), batch >= 16
    assert fwd(8, 16, 64, 64) == "direct3h (8 x 8 tiles)"          # layer1 of a 64^3 crop (round 4), from 128 workgroups on
    assert fwd(16, 12, 64, 64).startswith("implicit GEMM")         # layer1 of a 48^3 crop: a 12 x 12 plane fills 56 % of four tiles (measured slower)
    assert fwd(8, 14, 64, 64) == "direct3h (8 x 8 tiles)"          # ragged planes from 75 % fill on (round 5)
    # stride-2 block fronts (conv + shortcut in one launch): the 32^3 crops' two and layer3.0 of the 64^3 crops (round 6); layer2.0 of the
    # 64^3 crops (16^3 -> 8^3: a sample no longer fits a workgroup's LDS) stays on the generic launches
    assert lib.mi_conv3d_s2_fwd_usable(32, 8, 64, 128) == 1 and lib.mi_conv3d_s2_dgrad_usable(32, 8, 64, 128) == 1
    assert lib.mi_conv3d_s2_fwd_usable(32, 4, 128, 256) == 1 and lib.mi_conv3d_s2_dgrad_usable(32, 4, 128, 256) == 1
    assert lib.mi_conv3d_s2_fwd_usable(32, 8, 128, 256) == 1 and lib.mi_conv3d_s2_dgrad_usable(32, 8, 128, 256) == 1
    assert lib.mi_conv3d_s2_fwd_usable(32, 16, 64, 128) == 0 and lib.mi_conv3d_s2_dgrad_usable(32, 16, 64, 128) == 0
    # weight gradients (round 5: layer2's shape has direct3_wgrad_kernel<true> too, opt-in)
    def wgrad(n, d, ci, co, k=3, s=1, p=1):
        x = torch.randn(n, d, d, d, ci, device="cuda")
        do = (d + 2 * p - k) // s + 1
        dy = torch.randn(n, do, do, do, co, device="cuda")
        w = H.conv_weight_param(co, ci, k); w.data = w.data.cuda()
        H.conv_wgrad_into(x, dy, w, k, s, p)
        return lib.mi_debug_last_conv_kernel().decode()
    assert wgrad(4, 8, 64, 64) == "direct3_wgrad + reduce"
    assert wgrad(4, 4, 128, 128).startswith("implicit GEMM")       # (direct3_wgrad_kernel<true> is opt-in: slower inside the step)
    assert wgrad(4, 2, 256, 256).startswith("pair_wgrad")
    assert wgrad(16, 8, 128, 128) == "direct3_wgrad (128 channels) + reduce"     # layer2 of a 64^3 crop (round 5)
    assert wgrad(8, 16, 64, 64) == "direct3_wgrad (8 x 8 tiles) + reduce"         # layer1 of a 64^3 crop (round 5)
    assert wgrad(8, 14, 64, 64).startswith("implicit GEMM")                       # (ragged planes: forward / data gradient only)
    assert wgrad(16, 4, 256, 256).startswith("implicit GEMM")                     # layer3 of a 64^3 crop: the kernel exists, no gain (opt-in)
    assert wgrad(4, 4, 128, 64).startswith("implicit GEMM")
    monkeypatch.setenv("MI_D3S_WGRAD", "1")
    assert wgrad(4, 4, 128, 128) == "direct3s_wgrad + reduce"
    monkeypatch.delenv("MI_D3S_WGRAD")
    monkeypatch.setenv("MI_CONV_NO_DIRECT", "1")
    assert fwd(4, 8, 64, 64).startswith("implicit GEMM")
    assert wgrad(4, 4, 128, 128).startswith("implicit GEMM")


@pytest.mark.gpu
@pytest.mark.parametrize("m,ci,co,bias", [(64, 128, 128, False), (37, 256, 128, True)])
def test_linear_stats_epilogue_matches_statistics_pass(m, ci, co, bias):
    """mi_linear_stats_fwd_f32 (the SyncBN form of the Linear + BatchNorm fusion): the product equals the plain Linear's bit
    for bit and the column sums from its epilogue equal mi_bn_stats of that product (doubles over <= 64 rows: to rounding)."""
    import numpy as np
    from cet_pick_amd import hipops as H, _lib as L
    lib = L.lib()
    g = torch.Generator().manual_seed(m + ci)
    lin = H.HipLinear(ci, co, bias=bias).cuda()
    x = torch.randn(m, ci, generator=g).cuda()
    y_plain = lin(x).detach()
    y = torch.empty_like(y_plain)
    sums = torch.zeros(2 * co, dtype=torch.float64, device="cuda")
    w5 = H._as5(lin.weight)
    rc = lib.mi_linear_stats_fwd_f32(L.ptr(x), L.ptr(w5), L.ptr(lin.bias), L.ptr(y), L.ptr(sums), m, ci, co, L.stream())
    assert rc == 0
    assert torch.equal(y, y_plain)
    want = torch.cat([y.double().sum(0), (y.double() ** 2).sum(0)])
    scale = torch.cat([y.double().abs().sum(0), (y.double() ** 2).sum(0)])
    assert float(((sums - want).abs() / scale).max()) < 1e-12


@pytest.mark.gpu
def test_weight_gradients_of_a_backward_pass_that_raised_are_dropped():
    """layer1-shaped weight gradients wait for the END of a plain backward pass (one batched launch); a pass that raises never
    runs its final callbacks, and the queued jobs used to be launched into the NEXT pass's gradients.  They are now tagged with the
    autograd graph task and dropped by the first deferral of another pass."""
    from cet_pick_amd import hipops as H

    class Boom(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            return x.clone()

        @staticmethod
        def backward(ctx, g):
            raise RuntimeError("boom")

    g = torch.Generator().manual_seed(3)
    conv = H.HipConv3d(64, 64, 3, 1, 1).cuda()
    x = cl(torch.randn(4, 64, 8, 8, 8, generator=g))
    xin = x.clone().requires_grad_(True)
    y = conv(Boom.apply(xin))
    with pytest.raises(RuntimeError, match="boom"):
        (y * 3.0).sum().backward()                             # conv's weight gradient is queued, then the pass dies
    assert H._BACKWARD_END is not None and len(H._BACKWARD_END) == 1
    conv.weight.grad = None
    x2 = cl(torch.randn(4, 64, 8, 8, 8, generator=g)).requires_grad_(True)
    conv(x2).sum().backward()
    got = conv.weight.grad.clone()
    assert H._BACKWARD_END is None and not getattr(conv.weight, "_mi_wgrad_pending", False)
    conv.weight.grad = None
    conv(x2.detach().requires_grad_(True)).sum().backward()    # the same pass on a clean slate
    assert torch.equal(got, conv.weight.grad)
