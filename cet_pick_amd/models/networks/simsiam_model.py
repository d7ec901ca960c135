"""Mirror of the slice-wise SimSiam encoder `TomoResClassifier` (arch key 'simsiam', reference
models/networks/simsiam_model.py:159-236 constructor, :368-440 two-view forward, :322-366 forward_test,
`get_simsiam_net_small` :517-523): every z-slice of a sub-volume goes through a 2-D ResNet trunk
(conv7x7/s2 + BN + ReLU + maxpool 3/s2 + 3 BasicBlock stages), the slices are stacked back into a volume for one
Conv3d(256,256,3) + BatchNorm3d + ReLU, global average pool, fc 256 -> 256 and the proj / pred MLPs.

On channels-last storage the (b*d, h, w, C) output of the 2-D trunk IS the (b, d, h, w, C) volume: the reference's
reshape + permute pair (:402-409) costs nothing here.  Same parameter names / logical shapes as the reference.
"""
import torch
import torch.nn as nn

from ... import hipops as H
from .simsiam_model_2d import BN_MOMENTUM, BasicBlock, fill_fc_weights


class TomoResClassifier(nn.Module):
    def __init__(self, block, layers, heads, head_conv):
        self.inplanes = 64
        self.heads = heads
        self.deconv_with_bias = False
        super().__init__()
        self.conv1 = H.HipConv2d(1, 64, 7, stride=2, pad=3)
        self.bn1 = H.HipBatchNorm(64, momentum=BN_MOMENTUM)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.Identity()                       # MaxPool2d(3, stride 2, pad 1): mi_maxpool3d on D = 1
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
        self.layer3 = self._make_layer(block, 256, layers[2], stride=2)
        c = 256 * block.expansion
        self.feature_3d = nn.Sequential(H.HipConv3d(c, c, 3, stride=1, pad=1), H.HipBatchNorm(c, momentum=BN_MOMENTUM),
                                        nn.ReLU(inplace=True))
        with torch.no_grad():
            self.feature_3d[0].weight.normal_(std=0.001)   # fill_fc_weights on a Conv3d (:135-138)
        self.avgpool = nn.Identity()
        self.fc = H.HipLinear(c, 256)
        fill_fc_weights(self.fc)
        for head in self.heads:
            if "proj" in head:
                fc = nn.Sequential(H.HipLinear(256, 256, bias=False), H.HipBatchNorm(256), nn.ReLU(inplace=True),
                                   H.HipLinear(256, 256, bias=False), H.HipBatchNorm(256), nn.ReLU(inplace=True),
                                   H.HipLinear(256, 256, bias=False), H.HipBatchNorm(256, affine=False))
            if "pred" in head:
                fc = nn.Sequential(H.HipLinear(256, 256, bias=False), H.HipBatchNorm(256), nn.ReLU(inplace=True),
                                   H.HipLinear(256, 256))
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

    def _trunk(self, x1):
        if x1.dim() > 4:
            x1 = x1.squeeze(dim=1)
        b, d, h, w = x1.shape
        x = x1.contiguous().float().view(b * d, h, w, 1)                 # one image per slice
        x = self.bn1(self.conv1(x), relu=True)
        x = H.maxpool3d(x.unsqueeze(1), 3, 2, 1).squeeze(1)              # D = 1: the z window only sees the slice
        for layer in (self.layer1, self.layer2, self.layer3):
            for blk in layer:
                x = blk(x)
        _, hh, ww, ch = x.shape
        v = x.view(b, d, hh, ww, ch)                                     # slices are the z axis of the volume again
        v = self.feature_3d[1](self.feature_3d[0](v), relu=True)
        return self.fc(H.global_avgpool(v))

    def _proj(self, x):
        s = self.proj
        x = s[1](s[0](x), relu=True)
        x = s[4](s[3](x), relu=True)
        return s[7](s[6](x))

    def _pred(self, z):
        s = self.pred
        return s[3](s[1](s[0](z), relu=True))

    def forward_test(self, x1):
        z1 = self._proj(self._trunk(x1))
        ret1 = {}
        for head in self.heads:
            if "proj" in head:
                ret1[head] = z1.detach()
            if "pred" in head:
                ret1[head] = self._pred(z1)
        return ret1

    def forward(self, x1, x2):
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
        """:486-508 loads an ImageNet ResNet (file or URL) or raises; here None keeps the random init."""
        if local_path is None:
            return
        sd = torch.load(local_path, map_location="cpu")
        sd = sd.get("state_dict", sd)
        if "conv1.weight" in sd and sd["conv1.weight"].shape[1] == 3:
            sd["conv1.weight"] = sd["conv1.weight"].sum(dim=1, keepdim=True)
        own = self.state_dict()
        self.load_state_dict({k: v for k, v in sd.items() if k in own and own[k].shape == v.shape}, strict=False)


resnet_spec = {18: (BasicBlock, [2, 2, 2, 2]), 34: (BasicBlock, [3, 4, 6, 3])}


def get_simsiam_net_small(num_layers, heads, head_conv=32, last_k=0, local_path=None):
    block_class, layers = resnet_spec[num_layers]
    model = TomoResClassifier(block_class, layers, heads, head_conv=0)
    model.init_weights(num_layers, local_path=local_path)
    return model
