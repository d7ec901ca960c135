"""Detector network on the MI355X: mirror of `TomoConvUNet` (reference models/networks/unet_small.py:30-97) and
of the 2-D `UNet` it wraps (models/networks/unet.py: DownConv :198-249, UpConv :319-399, UNet :722-886, with the
options unet_small.py:38 fixes: dim=2, start_filts=32, merge 'concat', up 'transpose', BatchNorm, ReLU, 'same').

The module tree below only HOLDS the parameters - same names and shapes as the reference, so checkpoints are
interchangeable (tests/golden/ckpt_keys.json) - while `forward` runs the whole network in `libcetpick_hip.so`
on channels-last activations:
    per slice:  conv7x7/s2 + BN + ReLU -> [conv3x3 + BN + ReLU] x2 -> maxpool2 (ceil) ... -> transposed conv
                (1x1 implicit GEMM to 4*Co columns + pixel shuffle) + BN + ReLU -> concat -> convs ... -> conv1x1
    volume:     2 x Conv3d(3x3x3, dilation (1,4,4)) + ReLU -> heads Conv3d((3,1,1)); `proj` L2-normalised
This round builds the inference path (BatchNorm in eval mode, no autograd), which is what the detector
(detectors/tomo_det.py:23-37) runs; calling it in training mode raises.
"""
import math

import torch
import torch.nn as nn

from ... import _lib as L
from ... import hipops as H


class _DownConv(nn.Module):
    def __init__(self, ci, co, pooling):
        super().__init__()
        self.pooling = pooling
        self.conv1 = nn.Conv2d(ci, co, 3, padding=1, bias=False)
        self.conv2 = nn.Conv2d(co, co, 3, padding=1, bias=False)
        self.norm0 = nn.BatchNorm2d(co)
        self.norm1 = nn.BatchNorm2d(co)


class _UpConv(nn.Module):
    def __init__(self, ci, co):
        super().__init__()
        self.upconv = nn.ConvTranspose2d(ci, co, kernel_size=2, stride=2)
        self.conv1 = nn.Conv2d(2 * co, co, 3, padding=1, bias=False)
        self.conv2 = nn.Conv2d(co, co, 3, padding=1, bias=False)
        self.norm0 = nn.BatchNorm2d(co)
        self.norm1 = nn.BatchNorm2d(co)
        self.norm2 = nn.BatchNorm2d(co)


class UNet(nn.Module):
    """Parameter container of the 2-D U-Net (unet.py:722-886) for the configuration unet_small.py:38 uses."""

    def __init__(self, in_channels=1, out_channels=2, n_blocks=3, start_filts=32):
        super().__init__()
        if n_blocks < 1:
            raise ValueError("n_blocks must be > 1.")
        self.n_blocks = n_blocks
        self.down_convs = nn.ModuleList()
        self.up_convs = nn.ModuleList()
        outs = in_channels
        for i in range(n_blocks):
            ins, outs = outs, start_filts * (2 ** i)
            self.down_convs.append(_DownConv(ins, outs, pooling=i < n_blocks - 1))
        for i in range(n_blocks - 1):
            ins, outs = outs, outs // 2
            self.up_convs.append(_UpConv(ins, outs))
        self.conv_final = nn.Conv2d(outs, out_channels, kernel_size=1)
        for m in self.modules():             # unet.py:842-848
            if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
                nn.init.xavier_normal_(m.weight)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)


def _fill_fc_weights(layers):                # unet_small.py:16-28
    for m in layers.modules():
        if isinstance(m, (nn.Conv2d, nn.Conv3d)):
            nn.init.normal_(m.weight, std=0.001)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)


def _kernel_layout(w):
    """(Co, Ci, *k) logical weight -> a view with the same logical shape over [taps][Ci][Co] storage."""
    nd = w.dim() - 2
    perm = tuple(range(2, 2 + nd)) + (1, 0)
    phys = w.detach().permute(*perm).contiguous().float()
    inv = (nd + 1, nd) + tuple(range(nd))
    return phys.permute(*inv)


class TomoConvUNet(nn.Module):
    def __init__(self, n_blocks, heads, head_conv, last_k=3):
        super().__init__()
        self.heads = heads
        self.n_blocks = n_blocks
        self.conv1 = nn.Conv2d(1, 16, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(16)
        self.unet = UNet(16, out_channels=32, n_blocks=n_blocks)
        self.feature_head = nn.Sequential(
            nn.Conv3d(32, head_conv, kernel_size=(3, 3, 3), dilation=(1, 4, 4), padding=(1, 4, 4), bias=False),
            nn.ReLU(inplace=True),
            nn.Conv3d(head_conv, head_conv, kernel_size=(3, 3, 3), dilation=(1, 4, 4), padding=(1, 4, 4), bias=False),
            nn.ReLU(inplace=True))
        _fill_fc_weights(self.feature_head)
        for head in self.heads:
            fc = nn.Conv3d(head_conv, self.heads[head], kernel_size=(3, 1, 1), stride=1, padding=(1, 0, 0), bias=False)
            _fill_fc_weights(fc)
            self.__setattr__(head, fc)
        self._plan_key, self._plan_w = None, None

    # ---- kernel-layout copies of the weights, rebuilt when a parameter changes --------------------------
    def _weights(self):
        key = tuple((p.data_ptr(), p._version) for p in self.parameters())
        if key != self._plan_key:
            w = {}
            for name, m in self.named_modules():
                if isinstance(m, (nn.Conv2d, nn.Conv3d)):
                    w[name] = _kernel_layout(m.weight)
                elif isinstance(m, nn.ConvTranspose2d):
                    ci, co = m.weight.shape[:2]
                    # [Ci][(a*2+b)*Co + co] as a 1x1 convolution to 4*Co columns
                    phys = m.weight.detach().permute(0, 2, 3, 1).contiguous().float().view(1, 1, ci, 4 * co)
                    w[name] = phys.permute(3, 2, 0, 1)
            self._plan_key, self._plan_w = key, w
        return self._plan_w

    @staticmethod
    def _bn(x, bn, relu=True):
        c = x.shape[-1]
        m = x.numel() // c
        y = torch.empty_like(x)
        L.check(L.lib().mi_bn_eval_fwd(L.ptr(x), L.ptr(y), m, c, L.ptr(bn.running_mean), L.ptr(bn.running_var),
                                       L.ptr(bn.weight), L.ptr(bn.bias), bn.eps, None, None, int(relu), L.stream()),
                "mi_bn_eval_fwd")
        return y

    def _down(self, x, blk, w, prefix):
        y = self._bn(H.conv_fwd(x, w[prefix + ".conv1"], 3, 1, 1), blk.norm0)
        y = self._bn(H.conv_fwd(y, w[prefix + ".conv2"], 3, 1, 1), blk.norm1)
        if not blk.pooling:
            return y, y
        n, h, wd, c = y.shape
        p = torch.empty((n, (h + 1) // 2, (wd + 1) // 2, c), dtype=torch.float32, device=y.device)
        L.check(L.lib().mi_maxpool2d_ceil_fwd(L.ptr(y), L.ptr(p), None, n, h, wd, c, 2, L.stream()), "mi_maxpool2d_ceil_fwd")
        return p, y

    def _up(self, enc, dec, blk, w, prefix):
        n, h, wd, ci = dec.shape
        co = blk.upconv.weight.shape[1]
        t = H.conv_fwd(dec, w[prefix + ".upconv"], 1, 1, 0)                       # (n, h, w, 4*co)
        ho, wo = enc.shape[1], enc.shape[2]                                       # autocrop: odd encoder extents
        if not (2 * h - 1 <= ho <= 2 * h and 2 * wd - 1 <= wo <= 2 * wd):
            raise L.HipExtensionError("encoder / decoder extents do not match (%s vs 2x%s)" % (enc.shape, dec.shape))
        up = torch.empty((n, ho, wo, co), dtype=torch.float32, device=dec.device)
        lib = L.lib()
        L.check(lib.mi_shuffle2x2_fwd(L.ptr(t), L.ptr(blk.upconv.bias), L.ptr(up), n, h, wd, co, ho, wo, L.stream()),
                "mi_shuffle2x2_fwd")
        up = self._bn(up, blk.norm0)
        mrg = torch.empty((n, ho, wo, 2 * co), dtype=torch.float32, device=dec.device)
        L.check(lib.mi_concat_channels(L.ptr(up), co, L.ptr(enc), co, L.ptr(mrg), n * ho * wo, L.stream()),
                "mi_concat_channels")
        y = self._bn(H.conv_fwd(mrg, w[prefix + ".conv1"], 3, 1, 1), blk.norm1)
        return self._bn(H.conv_fwd(y, w[prefix + ".conv2"], 3, 1, 1), blk.norm2)

    def forward(self, x):
        if self.training or torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()) and x.requires_grad:
            raise NotImplementedError("TomoConvUNet on the MI355X is the inference path this round: call .eval() "
                                      "and run under torch.no_grad() (detector training, SURVEY.md C5, is not built)")
        if x.dim() > 4:
            x = x.squeeze()
        if x.dim() != 4:
            raise ValueError("expected (b, d, h, w) after squeeze, got %s" % (tuple(x.shape),))
        b, d, h, wd = x.shape
        x = L.require_cuda(x, "x").contiguous().view(b * d, h, wd, 1)             # one image per slice
        w = self._weights()
        lib = L.lib()
        y = self._bn(H.conv_fwd(x, w["conv1"], 7, 2, 3), self.bn1)
        skips = []
        for i, blk in enumerate(self.unet.down_convs):
            y, before = self._down(y, blk, w, "unet.down_convs.%d" % i)
            skips.append(before)
        for i, blk in enumerate(self.unet.up_convs):
            y = self._up(skips[-(i + 2)], y, blk, w, "unet.up_convs.%d" % i)
        y = H.conv_fwd(y, w["unet.conv_final"], 1, 1, 0)
        L.check(lib.mi_bias_add(L.ptr(y), L.ptr(self.unet.conv_final.bias), y.numel() // y.shape[-1], y.shape[-1],
                                L.stream()), "mi_bias_add")
        n, hh, ww, ch = y.shape
        v = y.view(b, d, hh, ww, ch)                                              # slices are the z axis again
        for name in ("feature_head.0", "feature_head.2"):
            v = H.conv_fwd(v, w[name], (3, 3, 3), 1, (1, 4, 4), relu=True, dil=(1, 4, 4))
        ret = {}
        for head in self.heads:
            k = self.heads[head]
            hc = v.shape[-1]
            if k <= 4:
                out = torch.empty((b, d, hh, ww, k), dtype=torch.float32, device=v.device)
                wk = getattr(self, head).weight.detach()[:, :, :, 0, 0].permute(2, 1, 0).contiguous().float()   # [3][C][K]
                L.check(lib.mi_zhead_fwd(L.ptr(v), L.ptr(wk), L.ptr(out), b, d, hh * ww, hc, k, L.stream()), "mi_zhead_fwd")
            else:
                out = H.conv_fwd(v, w[head], (3, 1, 1), 1, (1, 0, 0))
            if "proj" in head:
                flat = out.view(-1, k)
                o2 = torch.empty_like(flat)
                inv = torch.empty(flat.shape[0], dtype=torch.float32, device=flat.device)
                L.check(lib.mi_l2norm_fwd(L.ptr(flat), L.ptr(o2), L.ptr(inv), flat.shape[0], k, L.stream()), "mi_l2norm_fwd")
                out = o2.view(b, d, hh, ww, k)
            ret[head] = out.permute(0, 4, 1, 2, 3)                                # logical (B, C, D, H, W)
        return [ret]


def get_tomo_unet_small(num_layers, heads, head_conv, last_k=3, local_path=None):
    """unet_small.py `get_tomo_unet_small`: arch 'unet_<n_blocks>'."""
    return TomoConvUNet(num_layers, heads, head_conv, last_k)
