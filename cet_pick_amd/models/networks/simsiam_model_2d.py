"""Mirror of the SimSiam 2-D encoder of cet_pick/models/networks/simsiam_model_2d.py
(`BasicBlock` :473-502, `TomoResClassifier2D` :617-819, `get_simsiam2d_net_small` :928-932) on the
MI355X kernels: same constructor arguments, two-view `forward(x1, x2)`, `forward_test`, and the same
state_dict keys / logical shapes (conv weights (Cout,Cin,3,3), BatchNorm2d/1d, fc, proj, pred).

Not reproduced: `init_weights` needing an ImageNet ResNet-18 file or URL (the reference raises
RuntimeError without it, :847-876); here `local_path=None` keeps the random init and a given path
is loaded shape-tolerantly.
"""
import torch
import torch.nn as nn

from ... import hipops as H

BN_MOMENTUM = 0.1


def fill_fc_weights(layers):
    """simsiam_model_2d.py fill_fc_weights: N(0, 1e-3) linear weights, bias 0.001."""
    for m in layers.modules():
        if isinstance(m, H.HipLinear):
            nn.init.normal_(m.weight, std=0.001)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0.001)


class BasicBlock(nn.Module):
    """:473-502: conv3x3-BN-ReLU-conv3x3-BN, + residual (1x1 strided conv, no BN), ReLU."""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = H.HipConv2d(inplanes, planes, 3, stride=stride, pad=1)
        self.bn1 = H.HipBatchNorm(planes, momentum=BN_MOMENTUM)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = H.HipConv2d(planes, planes, 3, stride=1, pad=1)
        self.bn2 = H.HipBatchNorm(planes, momentum=BN_MOMENTUM)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        # The gradient x receives through the shortcut is added in conv1's data-gradient epilogue (hipops.GradSlot) instead of by an
        # element-wise launch of autograd's: an identity shortcut's comes from bn2's backward, a convolution shortcut's from that
        # convolution's own data gradient (created after conv1, so autograd runs it before conv1's backward).
        slot = H.grad_slot_for(x)
        out = self.bn1(self.conv1(x, grad_slot=slot), relu=True)
        out = self.conv2(out)
        if self.downsample is None:
            return self.bn2(out, relu=True, res=x, res_slot=slot)          # relu(bn2(out) + x)
        residual = self.downsample[0](x, dx_slot=slot)
        if len(self.downsample) > 1:                        # the 2d3d variant normalises the shortcut (:601-607 there)
            residual = self.downsample[1](residual)
        return self.bn2(out, relu=True, res=residual)       # relu(bn2(out) + residual)


class TomoResClassifier2D(nn.Module):
    def __init__(self, block, layers, heads, head_conv):
        self.inplanes = 64
        self.heads = heads
        self.deconv_with_bias = False
        super().__init__()
        self.conv1 = H.HipConv2d(1, 64, 3, stride=1, pad=1)
        self.bn1 = H.HipBatchNorm(64, momentum=BN_MOMENTUM)
        self.relu = nn.ReLU(inplace=True)
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
        self.layer3 = self._make_layer(block, 256, layers[2], stride=2)
        self.avgpool = nn.Identity()
        self.out_dim = head_conv
        self.fc = H.HipLinear(256 * block.expansion, self.out_dim)
        fill_fc_weights(self.fc)
        d = self.out_dim
        for head in self.heads:
            if "proj" in head:
                fc = nn.Sequential(H.HipLinear(d, d, bias=False), H.HipBatchNorm(d), nn.ReLU(inplace=True),
                                   H.HipLinear(d, d, bias=False), H.HipBatchNorm(d), nn.ReLU(inplace=True),
                                   H.HipLinear(d, d, bias=False), H.HipBatchNorm(d, affine=False))
            if "pred" in head:
                fc = nn.Sequential(H.HipLinear(d, d, bias=False), H.HipBatchNorm(d), nn.ReLU(inplace=True),
                                   H.HipLinear(d, d))
            fill_fc_weights(fc)
            self.__setattr__(head, fc)

    def _make_layer(self, block, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(H.HipConv2d(self.inplanes, planes * block.expansion, 1, stride=stride, pad=0))
        layers = [block(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.inplanes, planes))
        return nn.Sequential(*layers)

    # ---- channels-last trunk -------------------------------------------------------------------
    def _trunk(self, x1):
        if x1.dim() > 4:
            x1 = x1.squeeze(dim=1)
        b, c, h, w = x1.shape
        if c != 1:
            raise ValueError("the 2-D encoder takes single-channel patches (B,1,H,W)")
        x = x1.contiguous().float().view(b, h, w, 1)         # C == 1: NCHW is already channels-last
        x = self.bn1(self.conv1(x), relu=True)
        for layer in (self.layer1, self.layer2, self.layer3):
            for blk in layer:
                x = blk(x)
        x = H.global_avgpool(x)
        return self.fc(x)

    def _proj(self, x):
        s = self.proj
        x = s[1](s[0](x), relu=True)
        x = s[4](s[3](x), relu=True)
        return s[7](s[6](x))

    def _pred(self, z):
        s = self.pred
        return s[3](s[1](s[0](z), relu=True))

    def forward_test(self, x1):
        """:751-774: {'proj': z.detach(), 'pred': p}."""
        z1 = self._proj(self._trunk(x1))
        ret1 = {}
        for head in self.heads:
            if "proj" in head:
                ret1[head] = z1.detach()
            if "pred" in head:
                ret1[head] = self._pred(z1)
        return ret1

    def forward(self, x1, x2):
        """:776-819: both views through the same weights; 'proj' outputs are detached (SimSiam
        stop-gradient), 'pred' = pred_head(z) carries the gradient."""
        f1, f2 = self._trunk(x1), self._trunk(x2)
        z1, z2 = self._proj(f1), self._proj(f2)
        ret1, ret2 = {}, {}
        for head in self.heads:
            if "proj" in head:
                ret1[head], ret2[head] = z1.detach(), z2.detach()
            if "pred" in head:
                ret1[head], ret2[head] = self._pred(z1), self._pred(z2)
        return [ret1, ret2]

    def init_weights(self, num_layers, local_path=None):
        if local_path is None:
            return
        sd = torch.load(local_path, map_location="cpu")
        sd = sd.get("state_dict", sd)
        if "conv1.weight" in sd and sd["conv1.weight"].shape[1] == 3:
            sd["conv1.weight"] = sd["conv1.weight"].sum(dim=1, keepdim=True)
        own = self.state_dict()
        self.load_state_dict({k: v for k, v in sd.items() if k in own and own[k].shape == v.shape}, strict=False)


resnet_spec = {18: (BasicBlock, [2, 2, 2, 2]), 34: (BasicBlock, [3, 4, 6, 3])}


def get_simsiam2d_net_small(num_layers, heads, head_conv=32, last_k=0, local_path=None):
    block_class, layers = resnet_spec[num_layers]
    model = TomoResClassifier2D(block_class, layers, heads, head_conv=head_conv)
    model.init_weights(num_layers, local_path=local_path)
    return model
