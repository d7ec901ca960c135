"""GPU parity of the detector network (row a22) and of its glue kernels with torch fp32 / the reference outputs."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
HEADS = {"hm": 1, "proj": 32}


def _cl(x):       # NCHW / NCDHW -> channels-last contiguous cuda
    perm = (0,) + tuple(range(2, x.dim())) + (1,)
    return x.permute(*perm).contiguous().cuda()


def _cf(y):       # channels-last cuda -> NC... cpu
    perm = (0, y.dim() - 1) + tuple(range(1, y.dim() - 1))
    return y.permute(*perm).cpu()


@pytest.mark.parametrize("shape,k,pad,dil", [((2, 6, 20, 24, 32), (3, 3, 3), (1, 4, 4), (1, 4, 4)),
                                             ((1, 5, 17, 19, 16), (3, 3, 3), (1, 2, 3), (1, 2, 3)),
                                             ((1, 4, 12, 12, 32), (3, 1, 1), (1, 0, 0), (1, 1, 1))])
def test_dilated_conv_fwd_dgrad_wgrad(shape, k, pad, dil):
    from cet_pick_amd import hipops as H, _lib as L
    n, d, h, w, ci = shape
    co = 32
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(n, ci, d, h, w, generator=g, requires_grad=True)
    wt = (torch.randn(co, ci, *k, generator=g) * 0.1).requires_grad_()
    y = F.conv3d(x, wt, padding=pad, dilation=dil)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    wk = wt.detach().permute(2, 3, 4, 1, 0).contiguous().cuda().permute(4, 3, 0, 1, 2)
    got = H.conv_fwd(_cl(x.detach()), wk, k, 1, pad, dil=dil)
    np.testing.assert_allclose(_cf(got).numpy(), y.detach().numpy(), rtol=1e-4, atol=1e-4)
    lib = L.lib()
    xc, dyc = _cl(x.detach()), _cl(dy)
    ws = L.workspace(lib.mi_convnd_dil_workspace_bytes(n, d, h, w, ci, co, *k, *pad, *dil), xc.device, "conv")
    dx = torch.empty_like(xc)
    L.check(lib.mi_convnd_dil_dgrad_f32(L.ptr(dyc), L.ptr(wk), L.ptr(dx), None, None, n, d, h, w, ci, co, *k, *pad, *dil,
                                        L.ptr(ws), ws.numel(), L.stream()), "dgrad")
    np.testing.assert_allclose(_cf(dx).numpy(), x.grad.numpy(), rtol=1e-4, atol=2e-4)
    dw = torch.empty(k[0], k[1], k[2], ci, co, device="cuda")
    L.check(lib.mi_convnd_dil_wgrad_f32(L.ptr(xc), L.ptr(dyc), L.ptr(dw), n, d, h, w, ci, co, *k, *pad, *dil,
                                        L.ptr(ws), ws.numel(), L.stream()), "wgrad")
    ref = wt.grad.permute(2, 3, 4, 1, 0).numpy()
    np.testing.assert_allclose(dw.cpu().numpy(), ref, rtol=1e-3, atol=1e-3 * np.abs(ref).max())


@pytest.mark.parametrize("shape,co,k,pad,dil", [((2, 1, 256, 256, 32), 32, (1, 3, 3), (0, 1, 1), None),      # U-Net 32 -> 32
                                                ((3, 1, 150, 151, 64), 32, (1, 3, 3), (0, 1, 1), None),      # ragged rows, 64 -> 32
                                                ((1, 16, 64, 65, 32), 32, (3, 3, 3), (1, 4, 4), (1, 4, 4)),  # the dilated head
                                                ((2, 1, 256, 256, 32), 16, (1, 3, 3), (0, 1, 1), None),      # 16 of the 32 columns
                                                ((1, 1, 300, 300, 32), 32, (1, 1, 1), (0, 0, 0), None)])     # 1 x 1
def test_narrow_tile_forward_matches_wide_tile_and_float64(shape, co, k, pad, dil, monkeypatch):
    """conv_igemm_kernel<FWD, 128, 32, 32> (round 4: forward convolutions with at most 32 output channels on 128 x 32 tiles) against
    the 64 x 64 tile it replaces (MI_CONV_NARROW=0) and float64: same cut, same six products, same slice order - the two tiles agree
    to rounding-order noise, and both hold the f32-equivalent bound; with the fused residual + ReLU epilogue as well
    (models/networks/unet_small.py:30-97)."""
    from cet_pick_amd import hipops as H, _lib as L
    n, d, h, w, ci = shape
    g = torch.Generator().manual_seed(sum(shape) + co)
    x = torch.randn(n, ci, d, h, w, generator=g)
    wt = torch.randn(co, ci, *k, generator=g) * (2.0 / (ci * k[0] * k[1] * k[2])) ** 0.5
    y64 = F.conv3d(x.double().cuda(), wt.double().cuda(), padding=pad, dilation=dil or 1)
    wk = wt.permute(2, 3, 4, 1, 0).contiguous().cuda().permute(4, 3, 0, 1, 2)
    xc = _cl(x)
    res = torch.randn(y64.shape, generator=g)
    outs = {}
    for narrow in ("1", "0"):
        monkeypatch.setenv("MI_CONV_NARROW", narrow)
        y = H.conv_fwd(xc, wk, k, 1, pad, dil=dil)
        outs[narrow] = (y, H.conv_fwd(xc, wk, k, 1, pad, _cl(res), True, dil=dil))
    ref = y64.permute(0, 2, 3, 4, 1)
    scale = float(ref.abs().max())
    e_n = float((outs["1"][0].double() - ref).abs().max())
    e_w = float((outs["0"][0].double() - ref).abs().max())
    assert e_n <= 2e-6 * scale and e_n <= 1.5 * e_w + 1e-7 * scale
    assert float((outs["1"][0] - outs["0"][0]).abs().max()) <= 1e-6 * scale
    ref2 = torch.relu(ref + _cl(res).double())
    assert float((outs["1"][1].double() - ref2).abs().max()) <= 2e-6 * max(scale, float(ref2.abs().max()))


@pytest.mark.parametrize("n,h,w,ho,wo,co,ce", [(2, 8, 8, 16, 16, 32, 32), (3, 5, 7, 9, 14, 64, 64), (1, 16, 16, 31, 31, 128, 128),
                                              (2, 4, 4, 8, 7, 32, 64),
                                              (2, 16, 16, 32, 32, 32, 32), (1, 8, 32, 15, 63, 64, 32), (2, 8, 16, 16, 31, 32, 64)])
def test_upconv_tail_equals_shuffle_batchnorm_relu_concat(n, h, w, ho, wo, co, ce):
    """mi_upconv_tail_fwd (inference: pixel shuffle + transposed-convolution bias + evaluation-mode BatchNorm + ReLU + concat with the
    encoder feature in ONE pass, unet.py:319-399) against the separate passes, odd (auto-cropped) extents included.  Round 5: where the
    1 x 1 product runs on conv_d32.hip (rows % 8, columns % 16, <= 128 input channels: the last three cases) the tail is the product's
    EPILOGUE (mi_conv_d32_upconv_fwd_f32 + mi_copy_channels_into) - same comparison, and the two-launch form (UPCONV_FUSED off) too."""
    from cet_pick_amd import hipops as H
    g = torch.Generator().manual_seed(n + h + wo + co)
    up = H.HipConvTranspose2x2(2 * co, co)
    bn = H.HipBatchNorm(co)
    with torch.no_grad():
        up.bias.copy_(torch.randn(co, generator=g) * 0.2)
        bn.weight.copy_(torch.rand(co, generator=g) + 0.5); bn.bias.copy_(torch.randn(co, generator=g) * 0.1)
        bn.running_mean.copy_(torch.randn(co, generator=g) * 0.2); bn.running_var.copy_(torch.rand(co, generator=g) + 0.5)
    up, bn = up.cuda(), bn.cuda().eval()
    dec = torch.randn(n, h, w, 2 * co, generator=g).cuda()
    enc = torch.randn(n, ho, wo, ce, generator=g).cuda()
    with torch.no_grad():
        fused = H.upconv_bn_relu_concat(up, bn, dec, enc)
        ref = H.concat_channels(bn(up(dec, ho, wo), relu=True), enc)
    assert fused.shape == ref.shape == (n, ho, wo, co + ce)
    assert torch.equal(fused[..., co:], ref[..., co:])
    assert float((fused - ref).abs().max()) <= 2e-6 * max(1.0, float(ref.abs().max()))
    in_epilogue = h % 8 == 0 and w % 16 == 0 and 2 * co <= 128
    assert (getattr(up.weight, "_mi_d32_up", None) is not None) == in_epilogue
    if in_epilogue:
        saved, H.UPCONV_FUSED = H.UPCONV_FUSED, False
        try:
            with torch.no_grad():
                two = H.upconv_bn_relu_concat(up, bn, dec, enc)
        finally:
            H.UPCONV_FUSED = saved
        assert torch.equal(two[..., co:], fused[..., co:])
        assert float((two - fused).abs().max()) <= 2e-6 * max(1.0, float(ref.abs().max()))


def test_skip_connections_written_into_the_concatenation_equal_the_copied_form():
    """Round 5: at inference a down-convolution block writes its skip connection straight into the concatenation buffer of the
    up-convolution block that consumes it (mi_conv_d32_fwd_pool_strided_f32) and that block's transposed convolution writes the other
    channels there (mi_conv_d32_upconv_fwd_f32): no concatenation pass.  Same kernels, same values: the U-Net's output is EQUAL BIT FOR BIT
    to the form with the copies (CONCAT_DIRECT off) and to the one with the separate pooling / up-convolution tail passes
    (unet.py:198-249,319-399,722-886)."""
    from cet_pick_amd import hipops as H
    from cet_pick_amd.models.networks.unet_small import UNet
    torch.manual_seed(3)
    net = UNet(16, out_channels=32, n_blocks=4).cuda().eval()
    for m in net.modules():
        if isinstance(m, H.HipBatchNorm):
            with torch.no_grad():
                m.running_mean.normal_(0, 0.2); m.running_var.uniform_(0.5, 1.5); m.weight.uniform_(0.5, 1.5); m.bias.normal_(0, 0.1)
    x = torch.randn(3, 64, 96, 16, device="cuda")
    outs = {}
    with torch.no_grad():
        for tag, flags in (("direct", {}), ("copied", {"CONCAT_DIRECT": False}),
                           ("separate", {"CONCAT_DIRECT": False, "POOL_FUSED": False, "UPCONV_FUSED": False})):
            saved = {k: getattr(H, k) for k in flags}
            for k, v in flags.items():
                setattr(H, k, v)
            try:
                outs[tag] = net(x).clone()
            finally:
                for k, v in saved.items():
                    setattr(H, k, v)
    assert torch.equal(outs["direct"], outs["copied"])
    assert float((outs["direct"] - outs["separate"]).abs().max()) <= 2e-6 * float(outs["separate"].abs().max())


def test_skip_into_concatenation_declines_a_buffer_of_2_gib():
    """the concatenation buffer is co_up + co channels wide; from 0x7fff0000 bytes on the kernels that write into it
    decline (32-bit byte offsets), so the predicate must send such a level through the dense skip + copy form instead of raising - a
    32-slice chunk of a 1024 x 1024 tomogram (32 x 512 x 512 x 64 floats = 2^31 bytes) is the first case.  The U-Net forward on that
    chunk runs and equals the copied form bit for bit."""
    from cet_pick_amd import hipops as H
    from cet_pick_amd.models.networks.unet_small import UNet
    torch.manual_seed(5)
    net = UNet(16, out_channels=32, n_blocks=4).cuda().eval()
    blk, up = net.down_convs[0], net.up_convs[-1]
    with torch.no_grad():
        y31 = torch.empty(31, 512, 512, 32, device="cuda")
        y32 = torch.empty(32, 512, 512, 32, device="cuda")
        assert H.skip_into_concat_ok(blk.conv2, blk.norm1, y31, up.upconv.co, up=up.upconv, up_bn=up.norm0)
        assert not H.skip_into_concat_ok(blk.conv2, blk.norm1, y32, up.upconv.co, up=up.upconv, up_bn=up.norm0)
        up.norm0.train()                                   # an up block whose BatchNorm cannot be folded declines as well
        assert not H.skip_into_concat_ok(blk.conv2, blk.norm1, y31, up.upconv.co, up=up.upconv, up_bn=up.norm0)
        up.norm0.eval()
        del y31, y32
        x = torch.randn(32, 512, 512, 16, device="cuda")
        direct = net(x)
        saved, H.CONCAT_DIRECT = H.CONCAT_DIRECT, False
        try:
            copied = net(x)
        finally:
            H.CONCAT_DIRECT = saved
        assert torch.equal(direct, copied)


@pytest.mark.parametrize("ci,co,h,w", [(16, 32, 32, 48), (32, 32, 16, 16), (32, 64, 32, 32), (64, 64, 16, 48), (64, 128, 16, 32), (128, 128, 16, 16)])
def test_down_convolution_pool_in_the_epilogue_equals_the_pooling_pass(ci, co, h, w):
    """mi_conv_d32_fwd_pool_f32 (round 5; unet.py:198-249): conv -> folded BatchNorm -> ReLU with the 2 x 2 max-pool as a second output of
    the same launch: the un-pooled tensor equals the plain launch's bit for bit, the pooled one equals MaxPool2d(2, ceil_mode) of it bit
    for bit (every form of the kernel: 32 columns, 64-column blocks, 16 x 16 and 16 x 8 tiles)."""
    from cet_pick_amd import hipops as H
    g = torch.Generator().manual_seed(ci + co + h)
    conv = H.HipConv2d(ci, co, 3, 1, 1).cuda()
    bn = H.HipBatchNorm(co)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(co, generator=g) + 0.5); bn.bias.copy_(torch.randn(co, generator=g) * 0.1)
        bn.running_mean.copy_(torch.randn(co, generator=g) * 0.2); bn.running_var.copy_(torch.rand(co, generator=g) + 0.5)
    bn = bn.cuda().eval()
    x = torch.randn(3, h, w, ci, generator=g).cuda()
    with torch.no_grad():
        y, pooled = H.conv_bn(conv, bn, x, relu=True, pool=True)
        y_plain = H.conv_bn(conv, bn, x, relu=True)
        saved, H.POOL_FUSED = H.POOL_FUSED, False
        try:
            y2, pooled2 = H.conv_bn(conv, bn, x, relu=True, pool=True)
        finally:
            H.POOL_FUSED = saved
    assert torch.equal(y, y_plain) and torch.equal(y2, y_plain)
    assert pooled.shape == (3, h // 2, w // 2, co)
    assert torch.equal(pooled, H.maxpool2d_ceil(y_plain, 2)) and torch.equal(pooled2, pooled)


@pytest.mark.parametrize("shape,k_hm", [((1, 6, 24, 40, 32), 1), ((2, 3, 9, 7, 32), 1), ((1, 1, 16, 16, 32), 3), ((1, 5, 8, 24, 64), 4)])
def test_detector_heads_in_one_pass_equal_the_separate_heads(shape, k_hm):
    """smallk_head_kernel (round 5; unet_small.py:86-97): `proj` = F.normalize(Conv3d(C, 32, (3,1,1))(v)) and `hm` = Conv3d(C, K,
    (3,1,1))(v) in ONE pass over the feature volume - the normalisation in the product's epilogue, hm from the fragments it loads -
    against the three separate kernels (short-reduction convolution, L2 normalise, z head) and float64; ragged row counts, one plane
    (only the centre tap exists), two samples (no tap crosses a sample)."""
    from cet_pick_amd import hipops as H
    n, d, h, w, c = shape
    g = torch.Generator().manual_seed(sum(shape) + k_hm)
    v = (torch.randn(n, d, h, w, c, generator=g) * torch.exp(torch.randn(n, d, h, w, 1, generator=g))).cuda()
    proj = H.HipConvNd(c, 32, (3, 1, 1), (1, 0, 0)).cuda()
    hm = H.HipZHead(c, k_hm).cuda()
    with torch.no_grad():
        proj.weight.copy_(torch.randn(proj.weight.shape, generator=g).cuda() * 0.2)
        hm.weight.copy_(torch.randn(hm.weight.shape, generator=g).cuda() * 0.1)
        pair = H.detector_heads_fused(v, proj, hm)
        assert pair is not None
        sep_p = H.l2_normalize(proj(v).view(-1, 32)).view(n, d, h, w, 32)
        sep_h = hm(v)
    x64 = v.double().permute(0, 4, 1, 2, 3)
    with torch.no_grad():
        p64 = F.normalize(F.conv3d(x64, proj.weight.double(), padding=(1, 0, 0)), dim=1).permute(0, 2, 3, 4, 1)
        h64 = F.conv3d(x64, hm.weight.double(), padding=(1, 0, 0)).permute(0, 2, 3, 4, 1)
    assert float((pair[0].double() - p64).abs().max()) <= 2e-6
    assert float((sep_p.double() - p64).abs().max()) <= 2e-6
    sc = float(h64.abs().max())
    assert float((pair[1].double() - h64).abs().max()) <= 2e-6 * sc
    assert float((sep_h.double() - h64).abs().max()) <= 2e-6 * sc


@pytest.mark.parametrize("shape,co,k,pad", [((3, 1, 37, 41, 32), 32, (1, 1, 1), (0, 0, 0)),      # last layer of the U-Net, ragged rows
                                            ((2, 1, 32, 32, 64), 64, (1, 1, 1), (0, 0, 0)),      # two column blocks per wave
                                            ((1, 1, 16, 24, 256), 64, (1, 1, 1), (0, 0, 0)),     # sixteen k-steps
                                            ((2, 1, 8, 8, 16), 32, (1, 1, 1), (0, 0, 0)),        # one k-step
                                            ((2, 5, 9, 11, 32), 32, (3, 1, 1), (1, 0, 0)),       # the (3, 1, 1) head, two samples
                                            ((1, 1, 6, 6, 48), 64, (3, 1, 1), (1, 0, 0))])      # one plane: only the centre tap exists
def test_short_reduction_kernel_matches_generic_kernel_and_float64(shape, co, k, pad, monkeypatch):
    """smallk_fwd_kernel (inference: 1 x 1 and (3, 1, 1) convolutions without the implicit GEMM's tile pipeline; unet.py:319-399,880-886,
    unet_small.py:86-97) against the generic kernel (hipops.SMALLK off) and float64, with and without the bias + ReLU epilogue."""
    monkeypatch.setenv("MI_NO_D32_1X1", "1")        # (round 5: 1 x 1 products take conv_d32's tile-resident form first; this test is conv_smallk's)
    from cet_pick_amd import hipops as H, _lib as L
    n, d, h, w, ci = shape
    g = torch.Generator().manual_seed(sum(shape) + co)
    x = torch.randn(n, ci, d, h, w, generator=g) * torch.exp(torch.randn(n, ci, d, h, w, generator=g))
    wt = torch.randn(co, ci, *k, generator=g) * (1.0 / (ci * k[0])) ** 0.5
    b = torch.randn(co, generator=g) * 0.3
    y64 = F.conv3d(x.double().cuda(), wt.double().cuda(), padding=pad).permute(0, 2, 3, 4, 1)
    wk = wt.permute(2, 3, 4, 1, 0).contiguous().cuda().permute(4, 3, 0, 1, 2)
    xc = _cl(x)
    scale = float(y64.abs().max())
    with torch.no_grad():
        y = H.conv_fwd(xc, wk, k, 1, pad, inference=True)      # (the caller states inference: hipops.inference_mode())
        assert getattr(wk, "_mi_smallk", None) is not None              # the short-reduction path ran and kept its image
        yb = H.conv_bias_fwd(xc, wk, b.cuda(), k, 1, pad, relu=True)
        monkeypatch.setattr(H, "SMALLK", False)
        y_gen = H.conv_fwd(xc, wk, k, 1, pad, inference=True)
        yb_gen = H.conv_bias_fwd(xc, wk, b.cuda(), k, 1, pad, relu=True)
    # the f32-equivalent bound against float64; and no worse than the generic kernel beyond the order of the additions (same
    # cut, same six products per k-step: 16 k-steps in another grouping measured 1.8x)
    e, e_gen = float((y.double() - y64).abs().max()), float((y_gen.double() - y64).abs().max())
    assert e <= 2e-6 * scale and e <= 2.5 * e_gen + 1e-7 * scale
    ref_b = torch.relu(y64 + b.double().cuda())
    assert float((yb.double() - ref_b).abs().max()) <= 2e-6 * max(scale, 1.0)
    assert float((yb - yb_gen).abs().max()) <= 1e-6 * max(scale, 1.0)
    # the image follows an in-place change of the weights
    with torch.no_grad():
        monkeypatch.setattr(H, "SMALLK", True)
        wk.mul_(2.0)
        y2 = H.conv_fwd(xc, wk, k, 1, pad)
    assert float((y2.double() - 2 * y64).abs().max()) <= 4e-6 * scale


@pytest.mark.parametrize("n,h,w", [(3, 64, 64), (2, 37, 51), (1, 512, 512), (5, 7, 9)])
def test_first_layer_direct_kernel_matches_torch_and_the_generic_kernel(n, h, w, monkeypatch):
    """stem2d_fwd_kernel (Conv2d(1, 16, 7, stride 2, padding 3) + bias + ReLU at inference, unet_small.py:35) against torch and against
    the implicit GEMM with the same bias epilogue (MI_NO_STEM2D=1), odd extents and partial tiles included."""
    from cet_pick_amd import hipops as H
    g = torch.Generator().manual_seed(n + h + w)
    x = torch.randn(n, 1, h, w, generator=g)
    wt = torch.randn(16, 1, 7, 7, generator=g) * 0.2
    b = torch.randn(16, generator=g) * 0.3
    ref = F.relu(F.conv2d(x, wt, b, stride=2, padding=3))
    wk = wt.permute(2, 3, 1, 0).contiguous().cuda().permute(3, 2, 0, 1)
    xc = x.permute(0, 2, 3, 1).contiguous().cuda()
    y = H.conv_bias_fwd(xc, wk, b.cuda(), 7, 2, 3, relu=True)
    monkeypatch.setenv("MI_NO_STEM2D", "1")
    y_gen = H.conv_bias_fwd(xc, wk, b.cuda(), 7, 2, 3, relu=True)
    scale = float(ref.abs().max())
    np.testing.assert_allclose(y.permute(0, 3, 1, 2).cpu().numpy(), ref.numpy(), rtol=0, atol=2e-6 * scale)
    np.testing.assert_allclose(y.cpu().numpy(), y_gen.cpu().numpy(), rtol=0, atol=2e-6 * scale)


@pytest.mark.parametrize("rows,c", [(4096, 32), (70001, 32), (5000, 16), (4099, 64)])
def test_l2norm_rows_of_few_channels_equal_the_wave_per_row_kernel(rows, c, monkeypatch):
    """l2norm_small_fwd_kernel (C / 4 lanes per row; the detector's projection head, unet_small.py:93-97 `F.normalize(proj, dim=1)`)
    against the wave-per-row kernel (MI_L2NORM_GENERIC=1): BIT-identical rows and norms (the butterfly's order of additions is kept),
    a zero row included, and both against torch."""
    from cet_pick_amd import hipops as H
    g = torch.Generator().manual_seed(rows + c)
    x = (torch.randn(rows, c, generator=g) * torch.exp(2 * torch.randn(rows, 1, generator=g)))
    x[7] = 0.0
    xc = x.cuda().requires_grad_(True)
    y_small = H.l2_normalize(xc)
    dy = torch.randn(rows, c, generator=g).cuda()
    (gx_small,) = torch.autograd.grad(y_small, xc, dy)
    monkeypatch.setenv("MI_L2NORM_GENERIC", "1")
    y_gen = H.l2_normalize(xc)
    (gx_gen,) = torch.autograd.grad(y_gen, xc, dy)
    assert torch.equal(y_small.detach(), y_gen.detach()) and torch.equal(gx_small, gx_gen)
    np.testing.assert_allclose(y_small.detach().cpu().numpy(), F.normalize(x, dim=1).numpy(), rtol=2e-6, atol=1e-7)


@pytest.mark.parametrize("h,w", [(16, 16), (13, 9), (1, 7)])
def test_maxpool2d_ceil(h, w):
    from cet_pick_amd import _lib as L
    g = torch.Generator().manual_seed(h * w)
    x = torch.randn(3, 8, h, w, generator=g, requires_grad=True)
    y = F.max_pool2d(x, 2, ceil_mode=True)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    xc = _cl(x.detach())
    ho, wo = (h + 1) // 2, (w + 1) // 2
    out = torch.empty(3, ho, wo, 8, device="cuda")
    arg = torch.empty(3, ho, wo, 8, dtype=torch.uint8, device="cuda")
    lib = L.lib()
    L.check(lib.mi_maxpool2d_ceil_fwd(L.ptr(xc), L.ptr(out), L.ptr(arg), 3, h, w, 8, 2, L.stream()), "fwd")
    np.testing.assert_array_equal(_cf(out).numpy(), y.detach().numpy())
    dx = torch.empty_like(xc)
    L.check(lib.mi_maxpool2d_ceil_bwd(L.ptr(_cl(dy)), L.ptr(arg), L.ptr(dx), 3, h, w, 8, 2, L.stream()), "bwd")
    np.testing.assert_array_equal(_cf(dx).numpy(), x.grad.numpy())


@pytest.mark.parametrize("h,w,ho,wo", [(6, 5, 12, 10), (6, 5, 11, 9)])
def test_conv_transpose_as_gemm_plus_shuffle(h, w, ho, wo):
    from cet_pick_amd import hipops as H, _lib as L
    g = torch.Generator().manual_seed(ho)
    ci, co = 32, 16
    x = torch.randn(2, ci, h, w, generator=g)
    wt = torch.randn(ci, co, 2, 2, generator=g) * 0.2
    b = torch.randn(co, generator=g)
    ref = F.conv_transpose2d(x, wt, b, stride=2)[:, :, :ho, :wo]
    wk = wt.permute(0, 2, 3, 1).contiguous().cuda().view(1, 1, ci, 4 * co).permute(3, 2, 0, 1)
    t = H.conv_fwd(_cl(x), wk, 1, 1, 0)
    out = torch.empty(2, ho, wo, co, device="cuda")
    lib = L.lib()
    L.check(lib.mi_shuffle2x2_fwd(L.ptr(t), L.ptr(b.cuda()), L.ptr(out), 2, h, w, co, ho, wo, L.stream()), "shuffle")
    np.testing.assert_allclose(_cf(out).numpy(), ref.numpy(), rtol=1e-4, atol=1e-4)
    # backward scatter is the exact inverse (zeros in the cropped rim)
    dt = torch.empty_like(t)
    L.check(lib.mi_shuffle2x2_bwd(L.ptr(out), L.ptr(dt), 2, h, w, co, ho, wo, L.stream()), "unshuffle")
    back = torch.empty_like(out)
    L.check(lib.mi_shuffle2x2_fwd(L.ptr(dt), None, L.ptr(back), 2, h, w, co, ho, wo, L.stream()), "shuffle")
    assert torch.equal(back, out)


def test_concat_split_and_zhead():
    from cet_pick_amd import _lib as L
    g = torch.Generator().manual_seed(3)
    a, b = torch.randn(50, 16, generator=g).cuda(), torch.randn(50, 32, generator=g).cuda()
    out = torch.empty(50, 48, device="cuda")
    lib = L.lib()
    L.check(lib.mi_concat_channels(L.ptr(a), 16, L.ptr(b), 32, L.ptr(out), 50, L.stream()), "concat")
    assert torch.equal(out, torch.cat((a, b), 1))
    da, db = torch.empty_like(a), torch.empty_like(b)
    L.check(lib.mi_split_channels(L.ptr(out), L.ptr(da), 16, L.ptr(db), 32, 50, L.stream()), "split")
    assert torch.equal(da, a) and torch.equal(db, b)
    for k in (1, 3):
        x = torch.randn(2, 32, 6, 9, 7, generator=g)
        wt = torch.randn(k, 32, 3, 1, 1, generator=g) * 0.2
        ref = F.conv3d(x, wt, padding=(1, 0, 0))
        y = torch.empty(2, 6, 9, 7, k, device="cuda")
        wk = wt[:, :, :, 0, 0].permute(2, 1, 0).contiguous().cuda()
        L.check(lib.mi_zhead_fwd(L.ptr(_cl(x)), L.ptr(wk), L.ptr(y), 2, 6, 63, 32, k, L.stream()), "zhead")
        np.testing.assert_allclose(_cf(y).numpy(), ref.numpy(), rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("c,k", [(32, 1), (32, 3), (16, 4), (64, 2)])
def test_zhead_with_a_lane_group_per_voxel_matches_torch_and_the_thread_per_voxel_kernel(c, k, monkeypatch):
    """zhead_rows_kernel (round 4: C / 4 lanes per voxel read a row as one coalesced line; the heat-map head of the detector,
    unet_small.py:86-97, from 4,096 voxels on) against torch and the thread-per-voxel kernel (MI_ZHEAD_GENERIC=1)."""
    from cet_pick_amd import _lib as L
    g = torch.Generator().manual_seed(c + k)
    n, d, h, w = 2, 5, 27, 31
    x = torch.randn(n, c, d, h, w, generator=g)
    wt = torch.randn(k, c, 3, 1, 1, generator=g) * 0.2
    ref = F.conv3d(x, wt, padding=(1, 0, 0))
    wk = wt[:, :, :, 0, 0].permute(2, 1, 0).contiguous().cuda()
    xc = _cl(x)
    lib = L.lib()
    ys = []
    for generic in (False, True):
        if generic:
            monkeypatch.setenv("MI_ZHEAD_GENERIC", "1")
        y = torch.empty(n, d, h, w, k, device="cuda")
        L.check(lib.mi_zhead_fwd(L.ptr(xc), L.ptr(wk), L.ptr(y), n, d, h * w, c, k, L.stream()), "zhead")
        ys.append(y)
    np.testing.assert_allclose(_cf(ys[0]).numpy(), ref.numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(ys[0].cpu().numpy(), ys[1].cpu().numpy(), rtol=0, atol=2e-6 * float(ref.abs().max()))


def _net():
    from cet_pick_amd.models.networks.unet_small import TomoConvUNet
    from cet_pick_amd.synthetic import seeded_state_dict
    net = TomoConvUNet(4, HEADS, 32, 3)
    net.load_state_dict(seeded_state_dict(net, seed=321))
    return net.cuda().eval()


def test_unet_forward_matches_reference_outputs():
    g = np.load(os.path.join(G, "unet4.npz"))
    net = _net()
    with torch.no_grad():
        for tag in ("a", "odd", "b2"):
            out = net(torch.from_numpy(g[f"x_{tag}"]).cuda())[0]
            hm, pr = out["hm"].cpu().numpy(), out["proj"].cpu().numpy()
            ref = g[f"hm_{tag}"]
            assert hm.shape == ref.shape
            np.testing.assert_allclose(hm, ref, rtol=0, atol=2e-4 * np.abs(ref).max())
            np.testing.assert_allclose(pr[:, :, :, ::3, ::3], g[f"proj_{tag}"], rtol=0, atol=2e-4)


def test_unet_forward_vs_oracle_larger():
    from oracle import unet_ref as O
    net = _net()
    x = torch.randn(1, 12, 104, 120, generator=torch.Generator().manual_seed(12))
    sd = {k: v.cpu() for k, v in net.state_dict().items()}
    ref = O.tomo_conv_unet_forward(sd, x, 4, HEADS)
    with torch.no_grad():
        out = net(x.cuda())[0]
    for h in HEADS:
        r = ref[h].numpy()
        np.testing.assert_allclose(out[h].cpu().numpy(), r, rtol=0, atol=3e-4 * max(1.0, np.abs(r).max()))


def test_unet_head_in_z_slabs_equals_the_whole_volume():
    """Round 6 (found by the entry-point record on the 256 x 512 x 512 volume: the 32-channel feature map of such a volume is 2 GiB and the
    kernels address an operand with 32-bit byte offsets): the dilated 3-D head + both heads on z-slabs with a three-plane halo give the
    whole-volume outputs - same kernels on the same neighbourhoods, only a launch's tile origin moves: hm / proj to rounding order."""
    net = _net()
    x = torch.randn(1, 29, 64, 64, generator=torch.Generator().manual_seed(15)).cuda()
    with torch.no_grad():
        whole = {k: v.clone() for k, v in net(x)[0].items()}
        net.head_slab = 8                                 # 29 planes: slabs of 8, 8, 8 and 5 with halos
        try:
            slabbed = net(x)[0]
        finally:
            net.head_slab = 0
    for h in HEADS:
        assert slabbed[h].shape == whole[h].shape
        np.testing.assert_allclose(slabbed[h].cpu().numpy(), whole[h].cpu().numpy(), rtol=0, atol=2e-6 * max(1.0, float(whole[h].abs().max())))


def test_unet_inference_in_slice_chunks_matches_whole_volume_and_oracle():
    """Round 4: the evaluation-mode forward runs the per-slice 2-D U-Net `slice_chunk` slices at a time (ragged last chunk
    included) - same outputs as the whole volume at once (the only difference a split-K choice can make is rounding order),
    the oracle's values, and a peak of activations that no longer grows with the number of slices of the 2-D part."""
    from oracle import unet_ref as O
    net = _net()
    x = torch.randn(1, 37, 72, 88, generator=torch.Generator().manual_seed(14))
    xc = x.cuda()
    with torch.no_grad():
        net.slice_chunk = 0
        torch.cuda.reset_peak_memory_stats()
        base = torch.cuda.memory_allocated()
        whole = {k: v.clone() for k, v in net(xc)[0].items()}
        peak_whole = torch.cuda.max_memory_allocated() - base
        net.slice_chunk = 8
        torch.cuda.reset_peak_memory_stats()
        base = torch.cuda.memory_allocated()
        chunked = net(xc)[0]
        peak_chunk = torch.cuda.max_memory_allocated() - base
    ref = O.tomo_conv_unet_forward({k: v.cpu() for k, v in net.state_dict().items()}, x, 4, HEADS)
    for h in HEADS:
        np.testing.assert_allclose(chunked[h].cpu().numpy(), whole[h].cpu().numpy(), rtol=0, atol=2e-6 * max(1.0, float(whole[h].abs().max())))
        r = ref[h].numpy()
        np.testing.assert_allclose(chunked[h].cpu().numpy(), r, rtol=0, atol=3e-4 * max(1.0, np.abs(r).max()))
    assert peak_chunk < 0.6 * peak_whole, (peak_chunk, peak_whole)
    # training mode keeps the whole batch in one pass (batch-statistics BatchNorm)
    net.train()
    try:
        y = net(xc[:, :4])[0]["hm"]
        assert y.requires_grad
    finally:
        net.eval()


def test_unet_inference_with_folded_batchnorm_equals_the_unfolded_passes(monkeypatch):
    """Round 4: at inference evaluation-mode BatchNorm is folded into the convolution in front of it (scaled weights + a bias and
    ReLU epilogue: mi_convnd_fwd_bias_f32) and the last 1 x 1 convolution adds its bias and writes its chunk of the feature volume
    itself.  Against the unfolded sequence (conv, bn_apply, bias_add, chunk copy: hipops.FOLD_EVAL_BN off) on a network whose
    running statistics are not the identity, chunked and whole; and the folded weights follow an in-place change of the
    statistics (models/networks/unet.py:198-249,319-399, unet_small.py:30-97)."""
    from cet_pick_amd import hipops as H
    net = _net()
    g = torch.Generator().manual_seed(77)
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, H.HipBatchNorm):
                m.running_mean.copy_((torch.randn(m.running_mean.shape, generator=g) * 0.2).cuda())
                m.running_var.copy_((torch.rand(m.running_var.shape, generator=g) + 0.5).cuda())
                m.weight.copy_((torch.rand(m.weight.shape, generator=g) + 0.5).cuda())
                m.bias.copy_((torch.randn(m.bias.shape, generator=g) * 0.1).cuda())
        net.unet.conv_final.bias.copy_((torch.randn(net.unet.conv_final.bias.shape, generator=g) * 0.1).cuda())
    x = torch.randn(1, 21, 72, 88, generator=g).cuda()
    outs = {}
    with torch.no_grad():
        for fold in (True, False):
            monkeypatch.setattr(H, "FOLD_EVAL_BN", fold)
            for chunk in (8, 0):
                net.slice_chunk = chunk
                outs[(fold, chunk)] = {k: v.clone() for k, v in net(x)[0].items()}
    for h in HEADS:
        ref = outs[(False, 0)][h]
        tol = 2e-5 * max(1.0, float(ref.abs().max()))
        for key in ((True, 8), (True, 0), (False, 8)):
            assert float((outs[key][h] - ref).abs().max()) <= tol, (h, key)
    # the folded weights are rebuilt when a statistic changes in place
    monkeypatch.setattr(H, "FOLD_EVAL_BN", True)
    with torch.no_grad():
        net.bn1.running_var.mul_(4.0)
        a = net(x)[0]["hm"].clone()
        monkeypatch.setattr(H, "FOLD_EVAL_BN", False)
        b = net(x)[0]["hm"]
    assert float((a - outs[(True, 0)]["hm"]).abs().max()) > 0
    assert float((a - b).abs().max()) <= 2e-5 * max(1.0, float(b.abs().max()))


def test_unet_weight_cache_follows_parameter_updates():
    net = _net()
    x = torch.randn(1, 4, 32, 32, generator=torch.Generator().manual_seed(1)).cuda()
    with torch.no_grad():
        a = net(x)[0]["hm"].clone()
        net.hm.weight.mul_(2.0)
        b = net(x)[0]["hm"]
    np.testing.assert_allclose(b.cpu().numpy(), 2 * a.cpu().numpy(), rtol=1e-5, atol=1e-6)


def test_unet_training_forward_backward_vs_oracle():
    """train mode (batch-statistics BatchNorm) forward + backward through every layer type against torch autograd"""
    from oracle import unet_ref as O
    from cet_pick_amd.models.networks.unet_small import TomoConvUNet
    from cet_pick_amd.synthetic import seeded_state_dict
    net = TomoConvUNet(4, HEADS, 32, 3)
    sd0 = seeded_state_dict(net, seed=322)
    net.load_state_dict(sd0)
    net = net.cuda().train()
    g = torch.Generator().manual_seed(8)
    x = torch.randn(2, 4, 44, 52, generator=g)                      # odd extents below the first pool: autocrop path
    ref_sd = {k: v.clone().requires_grad_(v.is_floating_point() and not k.endswith(("running_mean", "running_var")))
              for k, v in sd0.items()}
    ref = O.tomo_conv_unet_forward(ref_sd, x, 4, HEADS, training=True)
    r1, r2 = torch.randn(ref["hm"].shape, generator=g), torch.randn(ref["proj"].shape, generator=g)
    ((ref["hm"] * r1).sum() + (ref["proj"] * r2).sum()).backward()
    # the same in float64: arbiter of the gradient comparison
    ref_sd64 = {k: (v.double() if v.is_floating_point() else v.clone()).clone().requires_grad_(
        v.is_floating_point() and not k.endswith(("running_mean", "running_var"))) for k, v in sd0.items()}
    ref64 = O.tomo_conv_unet_forward(ref_sd64, x.double(), 4, HEADS, training=True)
    ((ref64["hm"] * r1.double()).sum() + (ref64["proj"] * r2.double()).sum()).backward()
    from conftest import f32_equivalent
    out = net(x.cuda())[0]
    ((out["hm"] * r1.cuda()).sum() + (out["proj"] * r2.cuda()).sum()).backward()
    for h in HEADS:
        r = ref[h].detach().numpy()
        np.testing.assert_allclose(out[h].detach().cpu().numpy(), r, rtol=0, atol=3e-4 * max(1.0, np.abs(r).max()))
    worst = 0.0
    for name, prm in net.named_parameters():
        rg = ref_sd[name].grad
        assert prm.grad is not None and rg is not None, name
        a, b = prm.grad.detach().cpu().double(), rg.double()
        if name.endswith("upconv.bias"):
            # a bias in front of a batch-statistics BatchNorm: the gradient is exactly zero, both sides hold rounding noise
            wn = float(ref_sd[name.replace("bias", "weight")].grad.norm())
            assert float(a.norm()) < 1e-4 * wn and float(b.norm()) < 1e-4 * wn, name
            continue
        # deepest gradients (conv1, first blocks) accumulate fp32 rounding through ~25 layers with batch-statistics
        # BatchNorm: the GPU may sit as far from the float64 gradient as twice torch's own fp32 gradient does
        e_g, e_c = f32_equivalent(a.numpy(), b.numpy(), ref_sd64[name].grad.numpy(), floor=1e-5, what=name)
        worst = max(worst, e_g)
    # running statistics follow nn.BatchNorm2d's update
    np.testing.assert_allclose(net.bn1.running_var.cpu().numpy(), ref_sd["bn1.running_var"].numpy(), rtol=1e-4)
    assert int(net.bn1.num_batches_tracked) == 1


@pytest.mark.parametrize("case", ["2d_16", "2d_32", "2d_64", "head", "2d_32_to_64", "2d_64_to_64", "2d_128_to_64", "2d_64_to_128", "2d_128_to_128", "2d_128_to_256", "2d_256_to_128", "2d_256_to_256",
                                  "1x1_32_to_32", "1x1_64_to_128", "1x1_128_to_256", "1x1_256_to_512", "1x1_64_to_32"])
def test_direct_32_channel_kernel_matches_the_implicit_gemm_and_float64(case, monkeypatch):
    """conv_d32.hip (patch-resident direct convolution to 32 output channels, inference: the detector's 256 x 256 level and its
    dilated 3-D head) against the implicit GEMM it replaces (MI_NO_D32=1) and a float64 convolution: f32-equivalent (no further
    from float64 than the implicit GEMM by more than 2x), borders (zero padding, dilation classes at the image edge) included."""
    import torch.nn.functional as F
    from cet_pick_amd import hipops as H
    g = torch.Generator().manual_seed(17)
    co = int(case.split("_")[-1]) if "_to_" in case else 32      # (round 5: the same kernel to 64-column blocks of 64 / 128 / 256 channels)
    if case == "head":
        n, d, h, w, ci, k3, pad, dil = 1, 6, 64, 96, 32, (3, 3, 3), (1, 4, 4), (1, 4, 4)
    elif case.startswith("1x1"):          # (the transposed convolutions' products and the last layer: kind 4, the tile is the patch)
        monkeypatch.setenv("MI_D32_1X1_256", "1")          # (256 input channels: built, not faster than the implicit GEMM, opt-in)
        n, d, h, w, ci, k3, pad, dil = 3, 1, 40, 64, int(case.split("_")[1]), (1, 1, 1), (0, 0, 0), (1, 1, 1)
    else:
        n, d, h, w, ci, k3, pad, dil = 3, 1, 48, 64, int(case.split("_")[1]), (1, 3, 3), (0, 1, 1), (1, 1, 1)
    x = (torch.randn(n, d, h, w, ci, generator=g) * torch.exp(torch.randn(n, d, h, w, 1, generator=g))).cuda()
    wt = torch.randn(*k3, ci, co, generator=g) / (ci * k3[1] * k3[2]) ** 0.5          # kernel layout [kd, kh, kw, ci, co]
    wdev = wt.cuda().permute(4, 3, 0, 1, 2)                               # logical (co, ci, kd, kh, kw) over that storage
    bias = torch.randn(co, generator=g).cuda()
    want = F.conv3d(x.double().cpu().permute(0, 4, 1, 2, 3), wt.double().permute(4, 3, 0, 1, 2), bias.double().cpu(), padding=pad,
                    dilation=dil).clamp_min(0).permute(0, 2, 3, 4, 1)
    with torch.no_grad():
        if case == "head":
            want = F.conv3d(x.double().cpu().permute(0, 4, 1, 2, 3), wt.double().permute(4, 3, 0, 1, 2), None, padding=pad,
                            dilation=dil).clamp_min(0).permute(0, 2, 3, 4, 1)
            run = lambda: H.conv_fwd(x, wdev, k3, 1, pad, relu=True, dil=dil, inference=True)
        else:
            w4 = wdev[:, :, 0]                                                                      # 2-D layer: (co, ci, kh, kw)
            x4 = x[:, 0]
            run = lambda: H.conv_bias_fwd(x4, w4, bias, k3[1], 1, pad[1], relu=True).unsqueeze(1)
        got = run()
        kernel = H.L.lib().mi_conv_d32_kind(n, d, h, w, ci, co, *k3, *dil)
        assert kernel == (4 if case.startswith("1x1") else 3 if co >= 64 else 2 if case == "head" else 1)
        monkeypatch.setenv("MI_NO_D32", "1")
        ref = run()
        monkeypatch.delenv("MI_NO_D32")
    scale = float(want.abs().max())
    e_got = float((got.double().cpu() - want).abs().max()) / scale
    e_ref = float((ref.double().cpu() - want).abs().max()) / scale
    assert e_got <= 2 * e_ref + 1e-6, (e_got, e_ref)
    assert e_got < 5e-6
