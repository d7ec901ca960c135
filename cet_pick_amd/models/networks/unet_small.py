"""Detector network on the MI355X: mirror of `TomoConvUNet` (reference models/networks/unet_small.py:30-97) and of
the 2-D `UNet` it wraps (models/networks/unet.py: DownConv :198-249, UpConv :319-399, UNet :722-886, with the options
unet_small.py:38 fixes: dim=2, start_filts=32, merge 'concat', up 'transpose', BatchNorm, ReLU, 'same' convolutions).

Parameter names and logical shapes are the reference's (checkpoints are interchangeable, tests/golden/ckpt_keys.json);
weights are stored in the kernels' layouts and exposed as permuted views.  forward and backward run in
`libcetpick_hip.so` on channels-last activations:
    per slice:  conv7x7/s2 + BN + ReLU -> [conv3x3 + BN + ReLU] x2 -> maxpool2 (ceil) ... -> transposed conv
                (1x1 implicit GEMM to 4*Co columns + pixel shuffle) + BN + ReLU -> concat -> convs ... -> conv1x1
    volume:     2 x Conv3d(3x3x3, dilation (1,4,4)) + ReLU -> heads Conv3d((3,1,1)); `proj` L2-normalised
"""
import torch
import torch.nn as nn

from ... import _lib as L
from ... import hipops as H


def _xavier_normal_(w):
    """nn.init.xavier_normal_ on a kernel-layout parameter (fans come from the logical shape)."""
    rf = w[0][0].numel() if w.dim() > 2 else 1
    fan_in, fan_out = w.shape[1] * rf, w.shape[0] * rf
    with torch.no_grad():
        w.normal_(0.0, (2.0 / (fan_in + fan_out)) ** 0.5)


class DownConv(nn.Module):
    """unet.py:198-249: two 3x3 convolutions (+BN+ReLU) and a 2x2 ceil-mode max-pool."""

    def __init__(self, ci, co, pooling):
        super().__init__()
        self.pooling = pooling
        self.conv1 = H.HipConv2d(ci, co, 3, 1, 1)
        self.conv2 = H.HipConv2d(co, co, 3, 1, 1)
        self.norm0 = H.HipBatchNorm(co)
        self.norm1 = H.HipBatchNorm(co)

    def forward(self, x, co_up=None, up=None):
        """-> (input of the next level, skip connection, concatenation buffer or None).  co_up: the channels the up-convolution block
        that consumes the skip connection puts in front of it - at inference, where both layers run on the patch-resident kernel, the skip
        connection is written straight into that block's concatenation buffer (no concatenation pass)."""
        y = H.conv_bn(self.conv1, self.norm0, x, relu=True)
        if self.pooling:
            if co_up is not None and H.skip_into_concat_ok(self.conv2, self.norm1, y, co_up, up=up.upconv if up is not None else None,
                                                           up_bn=up.norm0 if up is not None else None):
                n, h, w, _ = y.shape
                cat = torch.empty((n, h, w, co_up + self.conv2.co), dtype=torch.float32, device=y.device)
                skip, pooled = H.conv_bn(self.conv2, self.norm1, y, relu=True, pool=True, out=cat[..., co_up:])
                return pooled, skip, cat
            y, pooled = H.conv_bn(self.conv2, self.norm1, y, relu=True, pool=True)     # (inference: the pool is conv2's epilogue)
            return pooled, y, None
        y = H.conv_bn(self.conv2, self.norm1, y, relu=True)
        return y, y, None


class UpConv(nn.Module):
    """unet.py:319-399: transposed conv (+BN+ReLU), concat with the encoder feature, two 3x3 convolutions."""

    def __init__(self, ci, co):
        super().__init__()
        self.upconv = H.HipConvTranspose2x2(ci, co)
        self.conv1 = H.HipConv2d(2 * co, co, 3, 1, 1)
        self.conv2 = H.HipConv2d(co, co, 3, 1, 1)
        self.norm0 = H.HipBatchNorm(co)
        self.norm1 = H.HipBatchNorm(co)
        self.norm2 = H.HipBatchNorm(co)

    def forward(self, enc, dec, cat=None):
        h, w = dec.shape[1], dec.shape[2]
        ho, wo = enc.shape[1], enc.shape[2]                  # autocrop (:253-266): odd encoder extents lose a row
        if not (2 * h - 1 <= ho <= 2 * h and 2 * w - 1 <= wo <= 2 * w):
            raise L.HipExtensionError("encoder / decoder extents do not match (%s vs 2x%s)" % (tuple(enc.shape), tuple(dec.shape)))
        y = H.conv_bn(self.conv1, self.norm1, H.upconv_bn_relu_concat(self.upconv, self.norm0, dec, enc, cat=cat), relu=True)
        return H.conv_bn(self.conv2, self.norm2, y, relu=True)


class UNet(nn.Module):
    """unet.py:722-886 for the configuration unet_small.py:38 uses."""

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
            self.down_convs.append(DownConv(ins, outs, pooling=i < n_blocks - 1))
        for i in range(n_blocks - 1):
            ins, outs = outs, outs // 2
            self.up_convs.append(UpConv(ins, outs))
        self.conv_final = H.HipConv2d(outs, out_channels, 1, 1, 0)
        self.conv_final.bias = nn.Parameter(torch.zeros(out_channels))
        for m in self.modules():             # unet.py:842-848: xavier-normal weights, zero biases
            if isinstance(m, (H.HipConv2d, H.HipConvTranspose2x2)):
                _xavier_normal_(m.weight)
                if getattr(m, "bias", None) is not None:
                    nn.init.constant_(m.bias, 0)

    def forward(self, x, out=None):
        """out (inference only): the tensor - e.g. a slice of the volume's feature map - that receives the result."""
        skips = []
        nd = len(self.down_convs)
        for j, blk in enumerate(self.down_convs):
            # the skip connection of down block j is consumed by up block nd - 2 - j
            up = self.up_convs[nd - 2 - j] if j < nd - 1 else None
            x, before_pool, cat = blk(x, co_up=up.upconv.co if up is not None else None, up=up)
            skips.append((before_pool, cat))
        for i, blk in enumerate(self.up_convs):
            enc, cat = skips[-(i + 2)]
            x = blk(enc, x, cat=cat)
        if not torch.is_grad_enabled() and x.is_cuda and H.FOLD_EVAL_BN:      # the 1 x 1 convolution with its bias in the epilogue
            return H.conv_bias_fwd(x, self.conv_final.weight, self.conv_final.bias, 1, 1, 0, out=out)
        y = H.bias_add(self.conv_final(x), self.conv_final)
        if out is not None:
            out.copy_(y)
            return out
        return y


class TomoConvUNet(nn.Module):
    def __init__(self, n_blocks, heads, head_conv, last_k=3):
        super().__init__()
        self.heads = heads
        self.n_blocks = n_blocks
        self.conv1 = H.HipConv2d(1, 16, 7, 2, 3)
        self.bn1 = H.HipBatchNorm(16)
        self.unet = UNet(16, out_channels=32, n_blocks=n_blocks)
        self.feature_head = nn.Sequential(H.HipConvNd(32, head_conv, (3, 3, 3), (1, 4, 4), (1, 4, 4)), nn.Identity(),
                                          H.HipConvNd(head_conv, head_conv, (3, 3, 3), (1, 4, 4), (1, 4, 4)), nn.Identity())
        for head, classes in self.heads.items():
            fc = H.HipZHead(head_conv, classes) if classes <= 4 else H.HipConvNd(head_conv, classes, (3, 1, 1), (1, 0, 0))
            self.__setattr__(head, fc)
        for m in list(self.feature_head) + [getattr(self, h) for h in self.heads]:   # fill_fc_weights, :16-28
            if hasattr(m, "weight"):
                with torch.no_grad():
                    m.weight.normal_(std=0.001)

    def _all_bn_eval(self):
        """Every BatchNorm under the net normalises with running statistics: only then are the slices independent and the
        chunked forward equal to the whole-volume one (a BatchNorm left in train mode, or built with
        track_running_stats=False, takes batch statistics - per chunk they would differ, and running statistics would be
        updated once per chunk)."""
        return all((not m.training) and m.track_running_stats and m.running_mean is not None
                   for m in self.modules() if isinstance(m, H.HipBatchNorm))

    def forward(self, x):
        if x.dim() > 4:
            x = x.squeeze()
        if x.dim() != 4:
            raise ValueError("expected (b, d, h, w) after squeeze, got %s" % (tuple(x.shape),))
        b, d, h, w = x.shape
        x = L.require_cuda(x, "x").contiguous().view(b * d, h, w, 1)              # one image per slice
        chunk = int(getattr(self, "slice_chunk", 16))
        if not self.training and not torch.is_grad_enabled() and chunk > 0 and b * d > chunk and self._all_bn_eval():
            # inference on a whole tomogram: the per-slice 2-D U-Net runs `slice_chunk` slices at a time (evaluation-mode
            # BatchNorm: slices are independent), so its activations - the skip connections of every level, 17.7 GB for a
            # 128 x 512 x 512 volume - exist for one chunk only; what stays is the 32-channel feature volume the 3-D head reads
            y = None
            for c0 in range(0, b * d, chunk):
                f = H.conv_bn(self.conv1, self.bn1, x[c0:c0 + chunk], relu=True)
                if y is None:                             # (the U-Net keeps its input's extent: 'same' convolutions, ceil-mode pools)
                    y = torch.empty((b * d,) + tuple(f.shape[1:3]) + (self.unet.conv_final.co,), dtype=f.dtype, device=f.device)
                self.unet(f, out=y[c0:c0 + chunk])        # the last convolution writes the chunk's slice itself
                del f
        else:
            y = self.unet(H.conv_bn(self.conv1, self.bn1, x, relu=True))
        _, hh, ww, ch = y.shape
        v = y.view(b, d, hh, ww, ch)                                              # slices are the z axis again
        slab = int(getattr(self, "head_slab", 0)) or (64 if 4 * v.numel() >= 0x7fff0000 else 0)
        if slab and b == 1 and d > slab and not self.training and not torch.is_grad_enabled():
            return [self._heads_in_z_slabs(v, slab)]
        return [self._heads(v)]

    def _heads_in_z_slabs(self, v, slab):
        """Inference on a volume whose 32-channel feature map reaches 2 GiB (256 x 512 x 512 in: 256 x 256 x 256 x 32 floats - the
        kernels address an operand with 32-bit byte offsets): the 3-D head runs on z-slabs with a halo of three planes - one per
        3 x 3 x 3 layer (z dilation 1) and one for the (3, 1, 1) heads -, whose interior planes are exactly the whole-volume result
        (at the volume's own ends the zero padding IS the reference's)."""
        b, d, hh, ww, _ = v.shape
        outs = {}
        for z0 in range(0, d, slab):
            lo, hi = max(0, z0 - 3), min(d, z0 + slab + 3)
            part = self._heads(v[:, lo:hi])
            n = min(slab, d - z0)
            for k, t in part.items():                                             # logical (B, C, D, H, W)
                if k not in outs:
                    outs[k] = torch.empty((b, d, hh, ww, t.shape[1]), dtype=t.dtype, device=t.device)
                outs[k][:, z0:z0 + n] = t.permute(0, 2, 3, 4, 1)[:, z0 - lo:z0 - lo + n]
            del part
        return {k: t.permute(0, 4, 1, 2, 3) for k, t in outs.items()}

    def _heads(self, v):
        b, d, hh, ww, ch = v.shape
        v = self.feature_head[0](v, relu=True)
        v = self.feature_head[2](v, relu=True)
        ret = {}
        if not self.training and not torch.is_grad_enabled() and set(self.heads) == {"hm", "proj"}:
            # inference: both heads in one pass over the feature volume (proj's normalisation in its epilogue, hm as a by-product)
            pair = H.detector_heads_fused(v, self.__getattr__("proj"), self.__getattr__("hm"))
            if pair is not None:
                ret["proj"] = pair[0].permute(0, 4, 1, 2, 3)
                ret["hm"] = pair[1].permute(0, 4, 1, 2, 3)
                return {k: ret[k] for k in self.heads}
        for head in self.heads:
            out = self.__getattr__(head)(v)
            if "proj" in head:
                k = out.shape[-1]
                out = H.l2_normalize(out.view(-1, k)).view(b, d, hh, ww, k)
            ret[head] = out.permute(0, 4, 1, 2, 3)                                # logical (B, C, D, H, W)
        return ret


def get_tomo_unet_small(num_layers, heads, head_conv, last_k=3, local_path=None):
    """unet_small.py `get_tomo_unet_small`: arch 'unet_<n_blocks>'."""
    return TomoConvUNet(num_layers, heads, head_conv, last_k)
