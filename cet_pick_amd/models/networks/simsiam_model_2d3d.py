"""Mirror of `TomoResClassifier2D3D` (arch key 'simsiam2d3d', reference models/networks/simsiam_model_2d3d.py:560-612
constructor, :733-790 two-view forward, :697-731 forward_test, `get_simsiam2d3d_net_small` :887-892): the tilt-series
patch and the tomogram patch of a particle share one 2-D ResNet trunk (they are stacked along the batch axis), their
pooled features are concatenated (2 x 256) and go through fc 512 -> head_conv and the proj / pred MLPs.

Built on the 2-D encoder mirror (simsiam_model_2d.py); the only new arithmetic is the (2B, 256) -> (B, 512) regrouping.
"""
import torch

import torch.nn as nn

from ... import hipops as H
from .simsiam_model_2d import BN_MOMENTUM, BasicBlock, TomoResClassifier2D, fill_fc_weights


class TomoResClassifier2D3D(TomoResClassifier2D):
    def __init__(self, block, layers, heads, head_conv):
        super().__init__(block, layers, heads, head_conv)
        self.fc = H.HipLinear(512 * block.expansion, self.out_dim)          # both modalities' features side by side
        fill_fc_weights(self.fc)

    def _make_layer(self, block, planes, blocks, stride=1):
        """:628-645: unlike the 2-D encoder, the 1x1 strided shortcut is followed by a BatchNorm2d."""
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(H.HipConv2d(self.inplanes, planes * block.expansion, 1, stride=stride, pad=0),
                                       H.HipBatchNorm(planes * block.expansion, momentum=BN_MOMENTUM))
        layers = [block(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.inplanes, planes))
        return nn.Sequential(*layers)

    def _trunk2(self, x_2d, x_3d):
        if x_2d.dim() > 4:
            x_2d = x_2d.squeeze(dim=1)
        x = torch.cat([x_2d, x_3d], dim=0)                                  # (2B, 1, h, w): one pass, shared BN statistics
        n, c, h, w = x.shape
        if c != 1:
            raise ValueError("the 2-D encoder takes single-channel patches (B,1,H,W)")
        x = x.contiguous().float().view(n, h, w, 1)
        x = self.bn1(self.conv1(x), relu=True)
        for layer in (self.layer1, self.layer2, self.layer3):
            for blk in layer:
                x = blk(x)
        f = H.global_avgpool(x)                                             # (2B, 256)
        f = torch.cat(torch.chunk(f, 2, dim=0), dim=1).contiguous()         # (B, 512): [tilt | tomogram]
        return self.fc(f)

    def forward_test(self, x1_2d, x1_3d):
        z1 = self._proj(self._trunk2(x1_2d, x1_3d))
        ret1 = {}
        for head in self.heads:
            if "proj" in head:
                ret1[head] = z1.detach()
            if "pred" in head:
                ret1[head] = self._pred(z1)
        return ret1

    def forward(self, x1_2d, x1_3d, x2_2d, x2_3d):
        f1, f2 = self._trunk2(x1_2d, x1_3d), self._trunk2(x2_2d, x2_3d)
        z1, z2 = self._proj(f1), self._proj(f2)
        ret1, ret2 = {}, {}
        for head in self.heads:
            if "proj" in head:
                ret1[head], ret2[head] = z1.detach(), z2.detach()
            if "pred" in head:
                ret1[head], ret2[head] = self._pred(z1), self._pred(z2)
        return [ret1, ret2]


resnet_spec = {18: (BasicBlock, [2, 2, 2, 2]), 34: (BasicBlock, [3, 4, 6, 3])}


def get_simsiam2d3d_net_small(num_layers, heads, head_conv=32, last_k=0, local_path=None):
    block_class, layers = resnet_spec[num_layers]
    model = TomoResClassifier2D3D(block_class, layers, heads, head_conv)
    model.init_weights(num_layers, local_path=local_path)
    return model
