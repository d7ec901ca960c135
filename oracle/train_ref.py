"""Oracle for the training path: plain torch fp32 on the CPU.

TEST INFRASTRUCTURE ONLY - see oracle/__init__.py.  A functional restatement (state_dict in,
tensors out) of the reference encoder / MoCo step, pinned by tests/golden/enc3d.npz and
tests/golden/moco_3steps.npz which were produced by the reference's own modules.
"""
import copy
import math

import numpy as np
import torch
import torch.nn.functional as F

BN_MOMENTUM = 0.1


def _bn(sd, prefix, x, train, momentum, affine=True):
    """nn.BatchNorm (train: batch stats + running-stat update in place in `sd`)."""
    w = sd.get(prefix + ".weight") if affine else None
    b = sd.get(prefix + ".bias") if affine else None
    return F.batch_norm(x, sd[prefix + ".running_mean"], sd[prefix + ".running_var"], w, b,
                        training=train, momentum=momentum, eps=1e-5)


def _relu(name, t, pre, masks):
    """ReLU `name`.  pre: optional dict that receives its input as name + ".pre".  masks: optional dict of boolean tensors -
    where `name` is in it the unit fires where the MASK says so (t * mask) instead of where t > 0: the same function wherever
    the two agree, and a chosen branch of the piecewise-linear network where a unit sits within rounding of zero (a test
    passes the decisions the GPU took, tests/test_trainer_gpu.py)."""
    if pre is not None:
        pre[name + ".pre"] = t
    if masks is not None and name in masks:
        return t * masks[name].to(t.dtype)
    return F.relu(t)


def _block(sd, prefix, x, stride, pre=None, masks=None):
    """moco_encoder_3d.py:55-84 BasicBlock (no BN)."""
    out = _relu(prefix + ".mid", F.conv3d(x, sd[prefix + ".conv1.weight"], stride=stride, padding=1), pre, masks)
    out = F.conv3d(out, sd[prefix + ".conv2.weight"], padding=1)
    res = x
    if prefix + ".downsample.0.weight" in sd:
        res = F.conv3d(x, sd[prefix + ".downsample.0.weight"], stride=stride)
    return _relu(prefix + ".out", out + res, pre, masks)


def encoder_forward(sd, x, train=True, acts=None, pre=None, relu_masks=None):
    """moco_encoder_3d.py:353-404 (`forward`) / :326-351 (`forward_test` when train=False).
    sd: dict name -> tensor with the reference's logical shapes; running stats are updated in
    place when train.  Returns proj (B,128).  pre / relu_masks: see `_relu` (names: layerL.B.mid, layerL.B.out, feature_3d,
    proj.1, proj.4; the stem's ReLU always takes its own decision)."""
    def rec(name, t):
        if acts is not None:
            acts[name] = t
        return t
    x = rec("conv1", F.conv3d(x, sd["conv1.weight"], stride=2, padding=3))
    x = rec("bn1", _bn(sd, "bn1", x, train, BN_MOMENTUM))
    x = F.relu(x)
    x = rec("maxpool", F.max_pool3d(x, 3, stride=2, padding=1))
    for li, stride in ((1, 1), (2, 2), (3, 2)):
        for bi in range(2):
            x = _block(sd, "layer%d.%d" % (li, bi), x, stride if bi == 0 else 1, pre, relu_masks)
        rec("layer%d" % li, x)
    x = F.conv3d(x, sd["feature_3d.0.weight"], padding=1)
    x = rec("feature_3d", _relu("feature_3d", _bn(sd, "feature_3d.1", x, train, BN_MOMENTUM), pre, relu_masks))
    x = F.adaptive_avg_pool3d(x, 1).reshape(x.shape[0], -1)
    x = rec("fc", F.linear(x, sd["fc.weight"], sd["fc.bias"]))
    # (`pre`: e.g. the BatchNorm outputs in front of the head's ReLUs - a test reads their distance from zero)
    x = _relu("proj.1", _bn(sd, "proj.1", F.linear(x, sd["proj.0.weight"]), train, 0.1), pre, relu_masks)
    x = _relu("proj.4", _bn(sd, "proj.4", F.linear(x, sd["proj.3.weight"]), train, 0.1), pre, relu_masks)
    x = _bn(sd, "proj.7", F.linear(x, sd["proj.6.weight"]), train, 0.1, affine=False)
    return x


PARAM_SUFFIX = (".weight", ".bias")


def param_names(sd):
    """Parameter (not buffer) names in module order, 'pred.*' aliases of 'proj.*' dropped."""
    return [k for k in sd if k.endswith(PARAM_SUFFIX) and not k.startswith("pred.")]


def moco_logits(q, k, queue, T):
    """models/moco.py:111-138."""
    l_pos = torch.einsum("nc,nc->n", [q, k]).unsqueeze(-1)
    l_neg = torch.einsum("nc,ck->nk", [q, queue.clone().detach()])
    return torch.cat([l_pos, l_neg], dim=1) / T


class MocoRef:
    """models/moco.py:12-146 + trains/tomo_moco_trainer.py:52,73 + SGD (moco_main.py:79) as one
    functional step over two state_dicts."""

    def __init__(self, sd_q, queue, m=0.999, T=0.1, lr=0.05):
        self.q = {k: v.clone() for k, v in sd_q.items()}
        self.k = {k: v.clone() for k, v in sd_q.items()}
        self.queue = queue.clone()
        self.ptr = 0
        self.m, self.T, self.lr = m, T, lr
        self.names = param_names(self.q)

    def step(self, im_q, im_k, pre=None, relu_masks=None):
        """pre: optional dict that receives the query encoder's ReLU inputs; relu_masks: the query encoder's ReLU decisions
        (encoder_forward).  The key encoder carries no gradient: a ReLU is continuous, its decisions do not matter there."""
        for n in self.names:
            self.q[n] = self.q[n].detach().requires_grad_(True)
        self.q.update({("pred" + n[4:]): self.q[n] for n in self.names if n.startswith("proj.")})
        qf = F.normalize(encoder_forward(self.q, im_q, True, None, pre, relu_masks), dim=1)
        with torch.no_grad():
            for n in self.names:                                    # moco.py:31-39
                self.k[n] = self.k[n] * self.m + self.q[n].detach() * (1.0 - self.m)
            kf = F.normalize(encoder_forward(self.k, im_k, True), dim=1)
        logits = moco_logits(qf, kf, self.queue, self.T)
        labels = torch.zeros(logits.shape[0], dtype=torch.long)
        b = kf.shape[0]
        assert self.queue.shape[1] % b == 0                          # moco.py:47
        self.queue[:, self.ptr:self.ptr + b] = kf.T                  # moco.py:49
        self.ptr = (self.ptr + b) % self.queue.shape[1]
        loss = F.cross_entropy(logits, labels)
        grads = torch.autograd.grad(loss, [self.q[n] for n in self.names])
        with torch.no_grad():
            for n, g in zip(self.names, grads):
                self.q[n] = (self.q[n] - self.lr * g).detach()
        return logits.detach(), float(loss.detach()), {n: g for n, g in zip(self.names, grads)}


def adjust_learning_rate(lr, epoch, lr_step, lr_decay_rate, cosine=False, num_epochs=None):
    """utils/utils.py:58-70."""
    if cosine:
        eta_min = lr * (lr_decay_rate ** 3)
        return eta_min + (lr - eta_min) * (1 + math.cos(math.pi * epoch / num_epochs)) / 2
    steps = int(np.sum(epoch > np.asarray(lr_step)))
    return lr * (lr_decay_rate ** steps) if steps > 0 else lr


# ------------------------------------------------------------------------------------------------
# a2 / a9: SimSiam 2-D encoder and loss
# ------------------------------------------------------------------------------------------------
def _block2d(sd, prefix, x, stride, train):
    """simsiam_model_2d.py:473-502 BasicBlock (BatchNorm2d inside, 1x1 strided downsample without BN)."""
    out = F.conv2d(x, sd[prefix + ".conv1.weight"], stride=stride, padding=1)
    out = F.relu(_bn(sd, prefix + ".bn1", out, train, BN_MOMENTUM))
    out = F.conv2d(out, sd[prefix + ".conv2.weight"], padding=1)
    out = _bn(sd, prefix + ".bn2", out, train, BN_MOMENTUM)
    res = x
    if prefix + ".downsample.0.weight" in sd:
        res = F.conv2d(x, sd[prefix + ".downsample.0.weight"], stride=stride)
        if prefix + ".downsample.1.weight" in sd:           # simsiam_model_2d3d.py: BatchNorm2d on the shortcut
            res = _bn(sd, prefix + ".downsample.1", res, train, BN_MOMENTUM)
    return F.relu(out + res)


def encoder2d_trunk(sd, x, train=True):
    """simsiam_model_2d.py:776-800 (one view): conv1-bn-relu, layer1-3, avgpool, fc."""
    x = F.relu(_bn(sd, "bn1", F.conv2d(x, sd["conv1.weight"], padding=1), train, BN_MOMENTUM))
    for li, stride in ((1, 1), (2, 2), (3, 2)):
        for bi in range(2):
            x = _block2d(sd, "layer%d.%d" % (li, bi), x, stride if bi == 0 else 1, train)
    x = F.adaptive_avg_pool2d(x, 1).reshape(x.shape[0], -1)
    return F.linear(x, sd["fc.weight"], sd["fc.bias"])


def simsiam_heads(sd, f, train=True):
    """proj (3 x Linear+BN, last affine=False) and pred (Linear-BN-ReLU-Linear) MLPs, :643-660."""
    z = F.relu(_bn(sd, "proj.1", F.linear(f, sd["proj.0.weight"]), train, 0.1))
    z = F.relu(_bn(sd, "proj.4", F.linear(z, sd["proj.3.weight"]), train, 0.1))
    z = _bn(sd, "proj.7", F.linear(z, sd["proj.6.weight"]), train, 0.1, affine=False)
    p = F.relu(_bn(sd, "pred.1", F.linear(z, sd["pred.0.weight"]), train, 0.1))
    p = F.linear(p, sd["pred.3.weight"], sd["pred.3.bias"])
    return z, p


def simsiam_forward(sd, x1, x2, train=True):
    """:776-819: views share weights; BN layers see view 1's trunk, then view 2's trunk, then the
    heads in the order z1, z2, p1, p2 (running statistics are updated in that order)."""
    f1 = encoder2d_trunk(sd, x1, train)
    f2 = encoder2d_trunk(sd, x2, train)
    # heads: proj(view1), proj(view2), pred(view1), pred(view2) - the reference's call order
    def proj(f):
        z = F.relu(_bn(sd, "proj.1", F.linear(f, sd["proj.0.weight"]), train, 0.1))
        z = F.relu(_bn(sd, "proj.4", F.linear(z, sd["proj.3.weight"]), train, 0.1))
        return _bn(sd, "proj.7", F.linear(z, sd["proj.6.weight"]), train, 0.1, affine=False)

    def pred(z):
        p = F.relu(_bn(sd, "pred.1", F.linear(z, sd["pred.0.weight"]), train, 0.1))
        return F.linear(p, sd["pred.3.weight"], sd["pred.3.bias"])
    z1, z2 = proj(f1), proj(f2)
    p1, p2 = pred(z1), pred(z2)
    return p1, z1, p2, z2


def simsiam_loss(p1, z1, p2, z2):
    """trains/tomo_simsiam_trainer.py:28-40."""
    cos = torch.nn.CosineSimilarity(dim=1)
    loss = -(cos(p1, z2.detach()).mean() + cos(p2, z1.detach()).mean()) * 0.5
    output_std = torch.std(F.normalize(p1.detach(), dim=1), 0).mean()
    return loss, output_std


# ------------------------------------------------------------------------------------------------
# a3: slice-wise SimSiam encoder (arch 'simsiam')
# ------------------------------------------------------------------------------------------------
def simsiam_slice_trunk(sd, x, train=True):
    """models/networks/simsiam_model.py:368-414 (one view): per-slice conv7x7/s2-bn-relu-maxpool, layer1-3,
    stack to a volume, feature_3d (Conv3d + BN3d + ReLU), avgpool, fc."""
    b, d, h, w = x.shape
    s = x.reshape(b * d, 1, h, w)
    s = F.relu(_bn(sd, "bn1", F.conv2d(s, sd["conv1.weight"], stride=2, padding=3), train, BN_MOMENTUM))
    s = F.max_pool2d(s, 3, 2, 1)
    for li, stride in ((1, 1), (2, 2), (3, 2)):
        for bi in range(2):
            s = _block2d(sd, "layer%d.%d" % (li, bi), s, stride if bi == 0 else 1, train)
    _, ch, hh, ww = s.shape
    v = s.reshape(b, d, ch, hh, ww).permute(0, 2, 1, 3, 4)
    v = F.relu(_bn(sd, "feature_3d.1", F.conv3d(v, sd["feature_3d.0.weight"], padding=1), train, BN_MOMENTUM))
    v = F.adaptive_avg_pool3d(v, 1).reshape(b, -1)
    return F.linear(v, sd["fc.weight"], sd["fc.bias"])


def simsiam_slice_forward(sd, x1, x2, train=True):
    f1 = simsiam_slice_trunk(sd, x1, train)
    f2 = simsiam_slice_trunk(sd, x2, train)

    def proj(f):
        z = F.relu(_bn(sd, "proj.1", F.linear(f, sd["proj.0.weight"]), train, 0.1))
        z = F.relu(_bn(sd, "proj.4", F.linear(z, sd["proj.3.weight"]), train, 0.1))
        return _bn(sd, "proj.7", F.linear(z, sd["proj.6.weight"]), train, 0.1, affine=False)

    def pred(z):
        p = F.relu(_bn(sd, "pred.1", F.linear(z, sd["pred.0.weight"]), train, 0.1))
        return F.linear(p, sd["pred.3.weight"], sd["pred.3.bias"])
    z1, z2 = proj(f1), proj(f2)
    return pred(z1), z1, pred(z2), z2


def simsiam2d3d_forward(sd, x1_2d, x1_3d, x2_2d, x2_3d, train=True):
    """models/networks/simsiam_model_2d3d.py:733-790: tilt and tomogram patches stacked along the batch through the
    shared 2-D trunk, pooled features regrouped (2B,256) -> (B,512), fc, proj / pred."""
    def trunk(a, b):
        x = torch.cat([a, b], 0)
        x = F.relu(_bn(sd, "bn1", F.conv2d(x, sd["conv1.weight"], padding=1), train, BN_MOMENTUM))
        for li, stride in ((1, 1), (2, 2), (3, 2)):
            for bi in range(2):
                x = _block2d(sd, "layer%d.%d" % (li, bi), x, stride if bi == 0 else 1, train)
        f = F.adaptive_avg_pool2d(x, 1).reshape(x.shape[0], -1)
        f = torch.cat(torch.chunk(f, 2, dim=0), dim=1)
        return F.linear(f, sd["fc.weight"], sd["fc.bias"])
    f1, f2 = trunk(x1_2d, x1_3d), trunk(x2_2d, x2_3d)

    def proj(f):
        z = F.relu(_bn(sd, "proj.1", F.linear(f, sd["proj.0.weight"]), train, 0.1))
        z = F.relu(_bn(sd, "proj.4", F.linear(z, sd["proj.3.weight"]), train, 0.1))
        return _bn(sd, "proj.7", F.linear(z, sd["proj.6.weight"]), train, 0.1, affine=False)

    def pred(z):
        p = F.relu(_bn(sd, "pred.1", F.linear(z, sd["pred.0.weight"]), train, 0.1))
        return F.linear(p, sd["pred.3.weight"], sd["pred.3.bias"])
    z1, z2 = proj(f1), proj(f2)
    return pred(z1), z1, pred(z2), z2


# ------------------------------------------------------------------------------------------------
# symmetric MoCo (trains/tomo_moco_small_trainer.py:24-161).  Pinned since round 3 by
# tests/golden/moco_small.npz: the reference's own MoCoModel.forward run on the CPU (its load-time imports of `progress`,
# `cv2`, `sknetwork`, `pytorch_metric_learning` are inert stubs in tests/golden/gen_golden.py::gen_moco_small).
# ------------------------------------------------------------------------------------------------
def symmetric_moco_step(sd_q, sd_k, queue, ptr, im1, im2, m, T, symmetric=True):
    """One forward of MoCoModel (tomo_moco_small_trainer.py:136-161) with moco3d encoders: returns loss, new key state
    dict, queue, ptr.  symmetric=False: the one-directional loss of :153-154 (only k(im2) is enqueued)."""
    names = param_names(sd_q)
    sd_k = dict(sd_k)
    for n in names:
        sd_k[n] = sd_k[n] * m + sd_q[n].detach() * (1.0 - m)

    def side(a, b):
        q = F.normalize(encoder_forward(sd_q, a, train=True), dim=1)
        with torch.no_grad():
            k = F.normalize(encoder_forward(sd_k, b, train=True), dim=1)
        logits = moco_logits(q, k, queue, T)
        return F.cross_entropy(logits, torch.zeros(logits.shape[0], dtype=torch.long)), k
    l12, k2 = side(im1, im2)
    if symmetric:
        l21, k1 = side(im2, im1)
        keys = torch.cat([k1, k2], 0)
    else:
        l21, keys = 0.0, k2
    queue = queue.clone()
    n = keys.shape[0]
    queue[:, ptr:ptr + n] = keys.t()
    return l12 + l21, sd_k, queue, (ptr + n) % queue.shape[1]
