"""Mirror of cet_pick/models/networks/moco_encoder_3d.py on the MI355X kernels.

Same class / factory names, constructor arguments, forward / forward_test outputs and state_dict
keys + logical shapes as the reference (moco_encoder_3d.py:156-236, 326-404, 470-478), so a
reference checkpoint loads into this module and vice versa.  Internally activations are
channels-last and every layer runs a hand-written HIP kernel (see cet_pick_amd/hipops.py).

Deliberate differences from the reference file (its defects listed in SURVEY.md §3.1 are not
reproduced): no per-step print(), `init_weights` does not read a hard-coded lab path, and the
factory accepts (and ignores) the `last_k` / `local_path` kwargs `create_model` passes.
"""
import torch
import torch.nn as nn

from ... import hipops as H

BN_MOMENTUM = 0.1


def fill_fc_weights(layers):
    """moco_encoder_3d.py:137-154: N(0, 1e-3) weights, conv/linear bias 0 / 0.001."""
    for m in layers.modules():
        if isinstance(m, H.HipConv3d):
            nn.init.normal_(m.weight, std=0.001)
        elif isinstance(m, H.HipLinear):
            nn.init.normal_(m.weight, std=0.001)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0.001)


class BasicBlock(nn.Module):
    """moco_encoder_3d.py:55-84 (the BatchNorms are commented out in the reference)."""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None):
        super().__init__()
        if dilation != 1:
            raise NotImplementedError("dilation != 1 is not used by the moco3d encoder")
        self.conv1 = H.HipConv3d(inplanes, planes, 3, stride=stride, pad=1)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = H.HipConv3d(planes, planes, 3, stride=1, pad=1)
        self.downsample = downsample
        self.stride = stride
        self.dilation = dilation

    def forward(self, x, mask_dx=False, dout_masked=False):
        return H.basic_block(x, self, mask_dx, dout_masked)


class TomoResClassifier3D(nn.Module):
    def __init__(self, block, layers, heads, head_conv):
        self.inplanes = 64
        self.heads = heads
        self.deconv_with_bias = False
        super().__init__()
        self.conv1 = H.HipConv3d(1, 64, 7, stride=2, pad=3)
        self.bn1 = H.HipBatchNorm(64, momentum=BN_MOMENTUM)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.Identity()            # placeholder: pooling runs as a kernel (k3, s2, p1)
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
        self.layer3 = self._make_layer(block, 256, layers[2], stride=2)
        self.feature_3d = nn.Sequential(
            H.HipConv3d(256 * block.expansion, 256 * block.expansion, 3, stride=1, pad=1),
            H.HipBatchNorm(256 * block.expansion, momentum=BN_MOMENTUM),
            nn.ReLU(inplace=True))
        fill_fc_weights(self.feature_3d)
        self.avgpool = nn.Identity()
        self.fc = H.HipLinear(256 * block.expansion, 128)
        fill_fc_weights(self.fc)
        for head in self.heads:
            if "proj" in head:
                fc = nn.Sequential(H.HipLinear(128, 128, bias=False), H.HipBatchNorm(128), nn.ReLU(inplace=True),
                                   H.HipLinear(128, 128, bias=False), H.HipBatchNorm(128), nn.ReLU(inplace=True),
                                   H.HipLinear(128, 128, bias=False), H.HipBatchNorm(128, affine=False))
            # 'pred' re-registers the module built for 'proj' (moco_encoder_3d.py:195-236: the pred
            # branch is commented out, so `fc` from the previous iteration is set again)
            fill_fc_weights(fc)
            self.__setattr__(head, fc)

    def _make_layer(self, block, planes, blocks, stride=1, dilation=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(H.HipConv3d(self.inplanes, planes * block.expansion, 1, stride=stride, pad=0))
        layers = [block(self.inplanes, planes, stride, dilation=dilation, downsample=downsample)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.inplanes, planes, dilation=dilation))
        return nn.Sequential(*layers)

    # Data-parallel training overlaps the gradient exchange with the backward pass: `grad_marker(tag)` (set by
    # MocoStepEngine) is called from autograd when the gradient of a stage boundary exists, i.e. when every
    # parameter gradient downstream of it has been enqueued.
    grad_marker = None

    def _mark(self, x, tag):
        if self.grad_marker is not None and x.requires_grad:
            x.register_hook(lambda g, t=tag: self.grad_marker(t))
        return x

    # ---- the trunk, channels-last -------------------------------------------------------------
    def _trunk(self, x1):
        b, c, d, h, w = x1.shape
        if c != 1:
            raise ValueError("the moco3d encoder takes single-channel sub-tomograms (B,1,D,H,W)")
        x = x1.contiguous().float().view(b, d, h, w, 1)      # C == 1: NCDHW is already channels-last
        # bn1's batch statistics come out of conv1's epilogue where the stem kernel runs (no pass over the 67 MB output)
        self.conv1.stats_for_bn = self.bn1.training or not self.bn1.track_running_stats
        x = self.conv1(x)
        sums = getattr(self.conv1, "bn_sums", None) if self.conv1.stats_for_bn else None
        self.conv1.bn_sums = None
        # bn1 + ReLU + MaxPool3d(3, 2, 1) as one fused layer: relu(bn(x)), the largest activation, is never stored
        H.stamp("stem")
        x = self._mark(H.bn_relu_maxpool3d(x, self.bn1, 3, 2, 1, sums=sums), "layer1")     # its gradient exists => layer1.. are done
        H.stamp("pool")
        # The ReLU at the end of a block is differentiated by the block's single consumer (the next block, then the
        # feature_3d convolution) in its data-gradient epilogue: six mask launches fewer per backward pass.
        first = True
        for tag, layer in (("layer2", self.layer1), ("layer3", self.layer2), (None, self.layer3)):
            for blk in layer:
                x = blk(x, mask_dx=not first, dout_masked=True)
                first = False
            if tag is not None:
                x = self._mark(x, tag)
            H.stamp("layer")
        x = self.feature_3d[0](x, mask_dx=True)
        if H.RELU_TAP is not None:
            # (the fused launch below never stores relu(bn(x)); the separate kernels give the same bits - on a copy of the
            # module, whose running statistics may move)
            import copy
            with torch.no_grad():
                H.RELU_TAP["feature_3d"] = copy.deepcopy(self.feature_3d[1])(x.detach(), relu=True).clone()
        x = H.bn_relu_global_avgpool(x, self.feature_3d[1])      # BatchNorm + ReLU + global average pool: one launch
        return self.fc(x)

    def _head(self, head, x):
        seq = self.__getattr__(head)
        H.stamp("trunk")
        x = H.linear_bn(x, seq[0], seq[1], relu=True)          # Linear + BatchNorm1d + ReLU: one launch each
        if H.RELU_TAP is not None:
            H.RELU_TAP[head + ".1"] = x.detach().clone()
        x = H.linear_bn(x, seq[3], seq[4], relu=True)
        if H.RELU_TAP is not None:
            H.RELU_TAP[head + ".4"] = x.detach().clone()
        return H.linear_bn(x, seq[6], seq[7])

    # ---- the same forward as a GENERATOR that stops in front of every SyncBN exchange (round 6) ---------------------
    def forward_sync_gen(self, x1):
        """`forward` under SyncBN (moco_main.py:64-66) with the five statistics all-reduces handed to the caller: yields this rank's
        column sums (2C doubles) of bn1, feature_3d.1, proj.1, proj.4, proj.7 in turn, expects them all-reduced IN PLACE when resumed,
        returns `forward`'s result.  MoCo drives encoder_q's and encoder_k's generators in lock step and all-reduces the two branches'
        sums of a layer as ONE collective (5 for the two forward passes of a step instead of 10).  Grad mode and stream are the caller's at
        every resume - none is entered in here (a context manager held across a yield would leak into the caller)."""
        b, c, d, h, w = x1.shape
        if c != 1:
            raise ValueError("the moco3d encoder takes single-channel sub-tomograms (B,1,D,H,W)")
        x = x1.contiguous().float().view(b, d, h, w, 1)
        self.conv1.stats_for_bn = True
        x = self.conv1(x)
        sums = getattr(self.conv1, "bn_sums", None)
        self.conv1.bn_sums = None
        if sums is None:
            sums = H.bn_local_sums(x)
        yield sums
        x = self._mark(H.bn_relu_maxpool3d(x, self.bn1, 3, 2, 1, sums=sums, reduced=True), "layer1")
        first = True
        for tag, layer in (("layer2", self.layer1), ("layer3", self.layer2), (None, self.layer3)):
            for blk in layer:
                x = blk(x, mask_dx=not first, dout_masked=True)
                first = False
            if tag is not None:
                x = self._mark(x, tag)
        x = self.feature_3d[0](x, mask_dx=True)
        sums = H.bn_local_sums(x)
        yield sums
        x = H.global_avgpool(self.feature_3d[1](x, relu=True, pre=("reduced", sums)))
        x = self.fc(x)
        ret1 = {}
        for head in self.heads:
            if "proj" in head:
                seq = self.__getattr__(head)
                y = x
                for li, bi, relu in ((0, 1, True), (3, 4, True), (6, 7, False)):
                    yl, sums = H.linear_with_local_sums(y, seq[li], seq[bi])
                    yield sums
                    y = seq[bi](yl, relu=relu, pre=("reduced", sums))
                ret1[head] = y
        return [ret1]

    def forward_test(self, x1):
        """moco_encoder_3d.py:326-351: {'proj': z.detach()}."""
        x = self._trunk(x1)
        ret1 = {}
        for head in self.heads:
            if "proj" in head:
                ret1[head] = self._head(head, x).detach()
        return ret1

    def forward(self, x1):
        """moco_encoder_3d.py:353-404: [{'proj': z}]."""
        x = self._trunk(x1)
        ret1 = {}
        for head in self.heads:
            if "proj" in head:
                ret1[head] = self._head(head, x)
        return [ret1]

    def init_weights(self, num_layers, local_path=None):
        """moco_encoder_3d.py:441-459 loads a ResNet-18 checkpoint from a fixed path; here the path
        is a parameter and a missing file leaves the random init (the reference would raise)."""
        if local_path is None:
            return
        ckpt = torch.load(local_path, map_location="cpu")
        sd = ckpt.get("state_dict", ckpt)
        sd = {(k[7:] if k.startswith("module") and not k.startswith("module_list") else k): v for k, v in sd.items()}
        if "conv1.weight" in sd and sd["conv1.weight"].shape[1] == 3:
            sd["conv1.weight"] = sd["conv1.weight"].sum(dim=1, keepdim=True)
        own = self.state_dict()
        sd = {k: v for k, v in sd.items() if k in own and own[k].shape == v.shape}
        self.load_state_dict(sd, strict=False)


resnet_spec = {18: (BasicBlock, [2, 2, 2, 2]),
               34: (BasicBlock, [3, 4, 6, 3])}


def get_moco_net_small_3d(num_layers, heads, head_conv=32, last_k=0, local_path=None):
    block_class, layers = resnet_spec[num_layers]
    model = TomoResClassifier3D(block_class, layers, heads, head_conv=0)
    model.init_weights(num_layers, local_path)
    return model
