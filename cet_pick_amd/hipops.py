"""torch.autograd glue over the training-path C-ABI (include/cetpick_hip.h).

Activations are channels-last fp32: 5-D (N, D, H, W, C) for volumes, 2-D (M, C) for vectors.
Weights are stored [tap][Cin][Cout]; modules expose them to state_dict() as permuted views with
the reference's logical shapes (Cout, Cin, kd, kh, kw) / (out, in).

Weight gradients are written straight into ``param.grad`` (a view of a flat gradient arena when
the model was flattened by :class:`ParamArena`): a parameter whose ``.grad`` is None gets the
gradient assigned, otherwise it is accumulated - the same contract as torch's AccumulateGrad.
PyTorch is the allocator / stream owner; all arithmetic happens in the HIP kernels.
"""
import os
import torch
import torch.nn as nn

from . import _lib as L


PROFILE = None          # bench.py: list collecting (tag, flops, start_event, end_event, launches) per conv call
PROFILE_REPEAT = 8      # timed back-to-back launches per call (hides the eager-mode gap in front of a lone launch)


def _prof_run(tag, flops, fn):
    """Run one conv call; under bench.py's roofline pass also time PROFILE_REPEAT more launches of the same call
    (idempotent: outputs never alias inputs) between two HIP events on the launch stream."""
    fn()
    if PROFILE is not None:
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(PROFILE_REPEAT):
            fn()
        e1.record()
        PROFILE.append((tag, flops, e0, e1, PROFILE_REPEAT))


STAMPS = None           # tools/stamp_step.py: (device int64 buffer, [names]) - wall-clock stamps recorded into the step


STAMP_TAG = ""          # prefix of the stamps' names (models/moco.py labels the two encoder branches)


def stamp(name):
    """Measurement aid: a one-thread launch on the current stream that writes the 100 MHz wall clock; a no-op unless
    tools/stamp_step.py armed it.  Recorded into a captured step it times the step's phases WITHOUT a profiler attached."""
    if STAMPS is None:
        return
    buf, names = STAMPS
    if len(names) >= buf.numel():
        raise L.HipExtensionError("stamp buffer full")
    L.check(L.lib().mi_debug_stamp(buf.data_ptr() + 8 * len(names), L.stream()), "mi_debug_stamp")
    names.append(STAMP_TAG + name)


def _ws(nbytes, device, tag):
    return L.workspace(max(int(nbytes), 256), device, tag)


def _f32c(t, name):
    L.require_cuda(t, name)
    if not t.is_contiguous():
        raise L.HipExtensionError(name + " must be contiguous (channels-last storage)")
    return t


# ------------------------------------------------------------------------------------------------
# parameter plumbing
# ------------------------------------------------------------------------------------------------
def conv_weight_param(co, ci, k, device=None):
    """Parameter with logical shape (co, ci, k, k, k) over physical storage [k,k,k,ci,co]."""
    phys = torch.empty(k, k, k, ci, co, device=device)
    return nn.Parameter(phys.permute(4, 3, 0, 1, 2))


def linear_weight_param(out_f, in_f, device=None):
    """Parameter with logical shape (out, in) over physical storage [in][out]."""
    phys = torch.empty(in_f, out_f, device=device)
    return nn.Parameter(phys.t())


def _phys_ok(p):
    """True when the parameter's strides are the kernel layout (co fastest, then ci, then taps)."""
    if p.dim() == 5:
        co, ci, kd, kh, kw = p.shape
        return p.stride() == (1, co, kh * kw * ci * co, kw * ci * co, ci * co)
    if p.dim() == 4:
        co, ci, kh, kw = p.shape
        return p.stride() == (1, co, kw * ci * co, ci * co)
    if p.dim() == 2:
        out_f, in_f = p.shape
        return p.stride() == (1, out_f)
    return p.is_contiguous()


def _dense_layout(p):
    """True when the tensor's strides are a permutation of a contiguous layout (every element of a numel-sized storage
    block is addressed exactly once): such a parameter can live in a flat arena under its own strides."""
    exp = 1
    for st, sz in sorted((st, sz) for st, sz in zip(p.stride(), p.shape) if sz != 1):
        if st != exp:
            return False
        exp *= sz
    return True


def _grad_target(param):
    """(tensor to write into, accumulate?) for a parameter's gradient."""
    g = param.grad
    if g is None:
        view = getattr(param, "_mi_grad_view", None)
        if view is None:
            view = torch.empty_like(param)        # preserve_format keeps the kernel layout
        param.grad = view
        return view, False
    view2 = getattr(param, "_mi_grad_view2", None)
    if view2 is not None and not param._mi_grad2_used:
        # the second contribution of a step (a two-view model: SimSiamStepEngine) goes into the second gradient arena - the optimizer
        # kernel adds the two - instead of through a temporary and a grad.add_ launch per parameter
        param._mi_grad2_used = True
        return view2, False
    if g.stride() != param.stride():
        raise L.HipExtensionError("param.grad layout differs from the parameter's kernel layout")
    return g, True


class ParamArena:
    """Flatten a module's parameters (and their gradients) into two contiguous fp32 arenas so the
    momentum update (models/moco.py:31-39), SGD (moco_main.py:79) and the gradient all-reduce are
    single passes.  Parameters keep their identity, logical shapes and kernel strides."""

    def __init__(self, module, second_grad_arena=False):
        params = [p for p in module.parameters()]
        if not params:
            raise ValueError("module has no parameters")
        dev = params[0].device
        offs, total = [], 0
        for p in params:
            offs.append(total)
            total += (p.numel() + 3) // 4 * 4          # keep every view 16-B aligned
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros(total, dtype=torch.float32, device=dev)
        self.params, self.offsets = params, offs
        for p, o in zip(params, offs):
            if not _dense_layout(p):
                raise L.HipExtensionError("parameter %s is not densely laid out" % (tuple(p.shape),))
            view = torch.as_strided(self.flat, p.shape, p.stride(), o)
            view.copy_(p.data)
            p.data = view
            p._mi_grad_view = torch.as_strided(self.flat_grad, p.shape, p.stride(), o)
            p.grad = None
        self.numel = total
        # a second gradient arena for models that apply every parameter twice per step (the two views of SimSiam): see _grad_target
        self.flat_grad2 = None
        if second_grad_arena:
            self.flat_grad2 = torch.zeros(total, dtype=torch.float32, device=dev)
            for p, o in zip(params, offs):
                p._mi_grad_view2 = torch.as_strided(self.flat_grad2, p.shape, p.stride(), o)
                p._mi_grad2_used = False

    def attach_grads(self):
        for p in self.params:
            p.grad = p._mi_grad_view

    def zero_grad(self):
        """set_to_none semantics: the next backward overwrites the arena views."""
        for p in self.params:
            p.grad = None
        if self.flat_grad2 is not None:
            for p in self.params:
                p._mi_grad2_used = False

    def settle_grads(self):
        """After a backward pass, before a kernel reads the flat arenas: a parameter the pass did not reach (or reached once, with two
        arenas) holds last step's values there - zero it.  (No launch when every parameter was reached, the usual case.)"""
        for p in self.params:
            if p.grad is None:
                p._mi_grad_view.zero_()
            elif p.grad.data_ptr() != p._mi_grad_view.data_ptr():
                p._mi_grad_view.copy_(p.grad)
            if self.flat_grad2 is not None and not p._mi_grad2_used:
                p._mi_grad_view2.zero_()


class GradExchange:
    """Data-parallel gradient exchange for a module trained with a stock optimizer (main.py:34-56, simsiam_main.py:
    the reference wraps the model in DistributedDataParallel): the parameters are re-homed in one flat fp32 arena, the
    backward pass writes their gradients into a second arena, and `sync()` - between `backward()` and
    `optimizer.step()` - averages that arena over the ranks with a few large asynchronous all-reduces (RCCL over xGMI:
    bucket = a quarter of the arena, >= 1 Mi floats) instead of one collective per tensor.  SyncBN statistics are
    exchanged by the BatchNorm modules themselves (convert_sync_batchnorm)."""

    def __init__(self, module, n_buckets=4):
        import torch.distributed as dist
        self.dist = dist
        self.world = dist.get_world_size()
        self.arena = ParamArena(module)
        n = self.arena.numel
        per = max((n + n_buckets - 1) // n_buckets, 1 << 20)
        per = (per + 3) // 4 * 4
        self.buckets = [(a, min(a + per, n)) for a in range(0, n, per)]
        self.calls = 0

    def sync(self):
        for prm in self.arena.params:
            view = prm._mi_grad_view
            if prm.grad is None:
                view.zero_()                       # a parameter the step did not reach contributes zeros
            elif prm.grad.data_ptr() != view.data_ptr():
                view.copy_(prm.grad)               # gradient produced outside the arena (plain autograd)
            prm.grad = view
        pending = [self.dist.all_reduce(self.arena.flat_grad[a:b], async_op=True) for a, b in self.buckets]
        for w in pending:
            w.wait()
        self.arena.flat_grad.mul_(1.0 / self.world)
        self.calls += 1

    def broadcast_parameters(self, src=0):
        """identical replicas before the first step (DistributedDataParallel does this at construction)"""
        self.dist.broadcast(self.arena.flat, src)


# ------------------------------------------------------------------------------------------------
# conv / linear
# ------------------------------------------------------------------------------------------------
def conv2d_weight_param(co, ci, k, device=None):
    """Parameter with logical shape (co, ci, k, k) over physical storage [k,k,ci,co]."""
    phys = torch.empty(k, k, ci, co, device=device)
    return nn.Parameter(phys.permute(3, 2, 0, 1))


def _k3(k, nd5):
    """int or tuple -> (kd, kh, kw); 2-D activations (N,H,W,C) use kd = 1."""
    if isinstance(k, (tuple, list)):
        return tuple(int(v) for v in k) if len(k) == 3 else (1, int(k[0]), int(k[1]))
    return (int(k),) * 3 if nd5 else (1, int(k), int(k))


def _p3(p, nd5):
    if isinstance(p, (tuple, list)):
        return tuple(int(v) for v in p) if len(p) == 3 else (0, int(p[0]), int(p[1]))
    return (int(p),) * 3 if nd5 else (0, int(p), int(p))


def _as5d(x):
    """(N,H,W,C) -> (N,1,H,W,C) view; 5-D passes through."""
    return x if x.dim() == 5 else x.unsqueeze(1)


def _out_dims(shape5, k3, stride, p3):
    n, d, h, w, _ = shape5
    return tuple((v + 2 * p - k) // stride + 1 for v, k, p in zip((d, h, w), k3, p3))


# Pre-cut weight images of the layer1-shaped convolutions (conv_direct3.hip).  Without a cache the C side rebuilds the image
# of a weight on every convolution call (one extra launch); an owner that knows when the weights change - MocoStepEngine:
# after the EMA kernel and after the SGD kernel - keeps them here and re-cuts a whole group in ONE launch.  The cache is
# only consulted while ACTIVE_IMAGES is set (the engine sets it around its step), so a stale image can never reach a
# convolution issued from anywhere else.
ACTIVE_IMAGES = None


class WeightImages:
    def __init__(self):
        self._groups = {}            # group -> list of (weight, dgrad flag, image tensor)
        self._s2_groups = {}         # group -> list of (weight, shortcut weight, dgrad flag, image tensor)
        self._by_weight = {}         # (weight storage address, dgrad [+ 2: stride-2 front]) -> image tensor

    @staticmethod
    def eligible(w, k, stride, pad):
        return (w.is_cuda and tuple(w.shape[:2]) in ((64, 64), (128, 128)) and _k3(k, True) == (3, 3, 3) and stride == 1
                and _p3(pad, True) == (1, 1, 1) and _phys_ok(w))

    def add(self, group, w, dgrad):
        nb = L.lib().mi_conv3d_direct_wimg_bytes(int(w.shape[0]))
        img = torch.empty(nb, dtype=torch.uint8, device=w.device)
        self._groups.setdefault(group, []).append((w, int(bool(dgrad)), img))
        self._by_weight[(w.data_ptr(), int(bool(dgrad)))] = img

    def get(self, w, dgrad, n, d, h, wd):
        img = self._by_weight.get((w.data_ptr(), int(bool(dgrad))))
        c = int(w.shape[0])
        if img is None or not L.lib().mi_conv3d_direct_usable(n, d, h, wd, c, c, 3, 1, 1):
            return None
        return img

    # ---- the stride-2 block fronts (csrc/conv_s2.hip): image of (conv1.weight, downsample weight), forward or data gradient ----
    def add_s2(self, group, w, w_ds, dgrad):
        lib = L.lib()
        co, ci = int(w.shape[0]), int(w.shape[1])
        nb = lib.mi_conv3d_s2_dgrad_workspace_bytes(ci, co) if dgrad else lib.mi_conv3d_s2_fwd_workspace_bytes(ci, co)
        img = torch.empty(int(nb), dtype=torch.uint8, device=w.device)
        self._s2_groups.setdefault(group, []).append((w, w_ds, int(bool(dgrad)), img))
        self._by_weight[(w.data_ptr(), 2 + int(bool(dgrad)))] = img

    def get_s2(self, w, dgrad):
        return self._by_weight.get((w.data_ptr(), 2 + int(bool(dgrad))))

    def _refresh_s2(self, group):
        items = self._s2_groups.get(group)
        if not items:
            return
        import ctypes
        n = len(items)
        ws = (ctypes.c_void_p * n)(*[it[0].data_ptr() for it in items])
        wd = (ctypes.c_void_p * n)(*[(it[1].data_ptr() if it[1] is not None else None) for it in items])
        imgs = (ctypes.c_void_p * n)(*[it[3].data_ptr() for it in items])
        ci = (ctypes.c_int * n)(*[int(it[0].shape[1]) for it in items])
        co = (ctypes.c_int * n)(*[int(it[0].shape[0]) for it in items])
        dg = (ctypes.c_int * n)(*[it[2] for it in items])
        cast = lambda a: ctypes.cast(a, ctypes.c_void_p)
        L.check(L.lib().mi_conv3d_s2_prep(cast(ws), cast(wd), cast(imgs), cast(ci), cast(co), cast(dg), n, L.stream()),
                "mi_conv3d_s2_prep")

    def refresh(self, group):
        """Re-cut every image of `group` from the current weights: one launch (per 16 images) on the current stream, and one
        for the group's stride-2 fronts."""
        self._refresh_s2(group)
        items = self._groups.get(group)
        if not items:
            return
        import ctypes
        n = len(items)
        ws = (ctypes.c_void_p * n)(*[it[0].data_ptr() for it in items])
        imgs = (ctypes.c_void_p * n)(*[it[2].data_ptr() for it in items])
        dg = (ctypes.c_int * n)(*[it[1] for it in items])
        ch = (ctypes.c_int * n)(*[int(it[0].shape[0]) for it in items])
        L.check(L.lib().mi_conv3d_direct_prep(ctypes.cast(ws, ctypes.c_void_p), ctypes.cast(imgs, ctypes.c_void_p),
                                              ctypes.cast(dg, ctypes.c_void_p), ctypes.cast(ch, ctypes.c_void_p), n,
                                              L.stream()), "mi_conv3d_direct_prep")

    def versions(self):
        """Sum of the torch version counters of the cached weights: changes when anything but the engine's own kernels
        (which refresh by themselves) wrote a weight."""
        return (sum(it[0]._version for items in self._groups.values() for it in items) +
                sum(it[0]._version + (it[1]._version if it[1] is not None else 0) for items in self._s2_groups.values() for it in items))


def _arith_bf16x3():
    """MI_CONV_ARITH (conv_igemm.hip conv_arith_bf16x3): anything starting with 'f' selects the f32 MFMA path."""
    import os
    return not os.environ.get("MI_CONV_ARITH", "").startswith("f")


def _cached_image(w, dgrad, n, d, h, wd, k3, stride, p3):
    if ACTIVE_IMAGES is None or PROFILE is not None or k3 != (3, 3, 3) or stride != 1 or p3 != (1, 1, 1):
        return None
    if not _arith_bf16x3():          # the direct kernels are bf16x3 only: an f32 A/B run keeps every conv on the f32 MFMA
        return None
    return ACTIVE_IMAGES.get(w, dgrad, n, d, h, wd)


SMALLK = os.environ.get("CETPICK_SMALLK", "1") != "0"

# Bumped by every kernel that writes parameters through raw pointers (sgd_step_, ema_update_: torch's version counters do
# not see those writes).  Part of the key of every cached derivative of a weight (conv_smallk images, folded BatchNorm).
WEIGHT_EPOCH = 0


def _bump_weight_epoch():
    global WEIGHT_EPOCH
    WEIGHT_EPOCH += 1


def inference_mode():
    """True when the CALLER runs without autograd.  To be evaluated where the user's grad mode is visible - a module's
    forward, never inside an autograd.Function (grad mode is always off in there) - and handed down explicitly."""
    return not torch.is_grad_enabled()


def _smallk_taps(x, w, k3, stride, p3, dil, nd5, inference):
    """1 / 3: this forward convolution can take the short-reduction inference kernel (conv_smallk.hip) as a 1 x 1 / (3, 1, 1)
    convolution; 0: not.  `inference`: the caller's word that no gradient will be asked of this call (inference_mode())."""
    if not SMALLK or not inference or not x.is_cuda or stride != 1:
        return 0
    if dil is not None and tuple(_k3(dil, nd5)) != (1, 1, 1):
        return 0
    ci, co = x.shape[-1], w.shape[0]
    # at most 64 output channels: every wave streams the whole weight image of its columns, which beyond that outweighs the
    # activations it reads (measured: the transposed convolutions' products to 128 - 512 columns are 20 % slower than on the implicit GEMM)
    if ci % 16 or co % 32 or co > 64 or x.numel() * 4 >= 0x7fff0000:
        return 0
    if tuple(k3) == (1, 1, 1) and tuple(p3) == (0, 0, 0):
        return 1
    if nd5 and tuple(k3) == (3, 1, 1) and tuple(p3) == (1, 0, 0) and 3 * ci <= 512:
        return 3
    return 0


def _smallk_image(w, owner, K, co):
    """The weight image of conv_smallk.hip for the kernel-layout weights `w` (a (K, co) matrix in memory), kept on `owner` (the
    parameter or folded-weight tensor that outlives the call) and rebuilt when the storage or its version changes."""
    lib = L.lib()
    holder = owner if owner is not None else w
    key = (w.data_ptr(), holder._version, WEIGHT_EPOCH, K, co)
    cache = getattr(holder, "_mi_smallk", None)
    if cache is None or cache[0] != key:
        img = torch.empty(int(lib.mi_smallk_image_bytes(K, co)), dtype=torch.uint8, device=w.device)
        L.check(lib.mi_smallk_prep(L.ptr(w), L.ptr(img), K, co, L.stream()), "mi_smallk_prep")
        cache = (key, img)
        try:
            holder._mi_smallk = cache
        except AttributeError:
            pass
    return cache[1]


def _d32_kind(x5shape, ci, co, k3, stride, p3, d3, inference):
    """0, or the kind (1: 3 x 3 per plane to 32 channels, 2: the dilated 3-D head, 3: 3 x 3 per plane to 64 channels) of conv_d32.hip's
    direct inference kernel for this forward convolution (padding = dilation * (k - 1) / 2)."""
    if not inference or stride != 1 or not (co == 32 or (co % 64 == 0 and 64 <= co <= 512)) or not _arith_bf16x3():
        return 0
    d3 = tuple(d3) if d3 is not None else (1, 1, 1)
    if tuple(p3) != tuple(dl * (kk - 1) // 2 for dl, kk in zip(d3, k3)):
        return 0
    n, d, h, wd, _ = x5shape
    return int(L.lib().mi_conv_d32_kind(n, d, h, wd, ci, co, *k3, *d3))


POOL_FUSED = os.environ.get("CETPICK_POOL_FUSED", "1") != "0"


def _d32_call(x, w, bias, relu, kind, owner=None, out=None, pool=False):
    """y = act(conv(x, w) + bias) on conv_d32.hip; the pre-cut weight image is kept on `owner` (rebuilt when the weights change)."""
    _f32c(x, "x")
    if not _phys_ok(w):
        raise L.HipExtensionError("conv weight is not in kernel layout [tap][Cin][Cout]")
    lib = L.lib()
    x5 = _as5d(x)
    n, d, h, wd, ci = x5.shape
    ntap = 27 if kind == 2 else 1 if kind == 4 else 9
    co = int(w.shape[0]) if kind in (3, 4) else 32
    holder = owner if owner is not None else w
    key = (w.data_ptr(), holder._version, WEIGHT_EPOCH, ci, ntap, co)
    cache = getattr(holder, "_mi_d32", None)
    if cache is None or cache[0] != key:
        if kind == 3 or (kind == 4 and co >= 64):
            img = torch.empty((co // 64) * int(lib.mi_conv_d64_image_bytes(ci, ntap)), dtype=torch.uint8, device=w.device)
            L.check(lib.mi_conv_d64_prep_co(L.ptr(w), L.ptr(img), ci, co, ntap, L.stream()), "mi_conv_d64_prep_co")
        else:
            img = torch.empty(int(lib.mi_conv_d32_image_bytes(ci, ntap)), dtype=torch.uint8, device=w.device)
            L.check(lib.mi_conv_d32_prep(L.ptr(w), L.ptr(img), ci, ntap, L.stream()), "mi_conv_d32_prep")
        cache = (key, img)
        try:
            holder._mi_d32 = cache
        except AttributeError:
            pass
    shape = tuple(x.shape[:-1]) + (co,)
    if pool:                                          # kinds 1 / 3: the 2 x 2 max-pool of the result as a second output
        cstride = co
        if out is None:
            out = torch.empty(shape, dtype=torch.float32, device=x.device)
        else:
            # a channel slice of a wider contiguous tensor (the concatenation buffer of the up-convolution block that will consume it)
            cstride = int(out.stride(-2))
            want = tuple(cstride * v for v in ([shape[-3] * shape[-2], shape[-2], 1] if len(shape) == 4 else
                                               [shape[-4] * shape[-3] * shape[-2], shape[-3] * shape[-2], shape[-2], 1])) + (1,)
            if (tuple(out.shape) != shape or out.dtype != torch.float32 or out.device != x.device or tuple(out.stride()) != want
                    or cstride < co or cstride % 4 or out.data_ptr() % 16):
                raise L.HipExtensionError("`out` must be an fp32 %s tensor or a channel slice of a wider contiguous one" % (shape,))
        pooled = torch.empty(tuple(x.shape[:-3]) + (h // 2, wd // 2, co), dtype=torch.float32, device=x.device)
        L.check(lib.mi_conv_d32_fwd_pool_strided_f32(L.ptr(x), L.ptr(cache[1]), L.ptr(bias), L.ptr(out), cstride, L.ptr(pooled),
                                                     int(relu), n, d, h, wd, ci, co, L.stream()), "mi_conv_d32_fwd_pool_strided_f32")
        return out, pooled
    if out is None:
        out = torch.empty(shape, dtype=torch.float32, device=x.device)
    elif tuple(out.shape) != shape or out.dtype != torch.float32 or not out.is_contiguous() or out.device != x.device:
        raise L.HipExtensionError("`out` must be a contiguous fp32 %s tensor on %s" % (shape, x.device))

    def call():
        if kind == 4:
            return L.check(lib.mi_conv_d32_1x1_fwd_f32(L.ptr(x), L.ptr(cache[1]), L.ptr(bias), L.ptr(out), int(relu), n, d, h, wd, ci, co,
                                                       L.stream()), "mi_conv_d32_1x1_fwd_f32")
        if kind == 3:
            return L.check(lib.mi_conv_d64_fwd_f32(L.ptr(x), L.ptr(cache[1]), L.ptr(bias), L.ptr(out), int(relu), n, d, h, wd, ci, co,
                                                   L.stream()), "mi_conv_d64_fwd_f32")
        return L.check(lib.mi_conv_d32_fwd_f32(L.ptr(x), L.ptr(cache[1]), L.ptr(bias), L.ptr(out), int(relu), n, d, h, wd, ci, kind,
                                               L.stream()), "mi_conv_d32_fwd_f32")
    _prof_run("fwd", 2.0 * n * d * h * wd * co * ci * ntap, call)
    return out


HEADS_FUSED = os.environ.get("CETPICK_HEADS_FUSED", "1") != "0"


def detector_heads_fused(v, proj, hm):
    """(normalize(proj(v)), hm(v)) for the detector's heads at inference in ONE pass over the feature volume v (N, D, H, W, C) - or None
    when the pair is not the one conv_smallk.hip's head kernel takes (proj: HipConvNd(C, 32, (3,1,1), padding (1,0,0)), hm: HipZHead(C,
    K <= 4); unet_small.py:86-97).  The caller has established inference (no gradient)."""
    if not (HEADS_FUSED and SMALLK and v.is_cuda and v.dim() == 5 and _arith_bf16x3()):
        return None
    if not (isinstance(proj, HipConvNd) and isinstance(hm, HipZHead)):
        return None
    n, d, h, wd, c = v.shape
    if (proj.co != 32 or proj.ci != c or tuple(proj.k) != (3, 1, 1) or tuple(proj.pad) != (1, 0, 0) or tuple(proj.dil) != (1, 1, 1)
            or hm.c != c or not 1 <= hm.k_out <= 4 or c % 16 or 3 * c > 512 or v.numel() * 4 >= 0x7fff0000
            or not _phys_ok(proj.weight) or not hm.weight.permute(2, 3, 4, 1, 0).is_contiguous()):
        return None
    _f32c(v, "v")
    img = _smallk_image(proj.weight, None, 3 * c, 32)
    m = n * d * h * wd
    y = torch.empty((n, d, h, wd, 32), dtype=torch.float32, device=v.device)
    yh = torch.empty((n, d, h, wd, hm.k_out), dtype=torch.float32, device=v.device)
    L.check(L.lib().mi_smallk_heads_fwd_f32(L.ptr(v), L.ptr(img), L.ptr(y), L.ptr(hm.weight), L.ptr(yh), hm.k_out, m, c, h * wd, d,
                                            L.stream()), "mi_smallk_heads_fwd_f32")
    return y, yh


def _smallk_call(x, w, bias, relu, ntaps, out=None, owner=None):
    _f32c(x, "x")
    if not _phys_ok(w):
        raise L.HipExtensionError("conv weight is not in kernel layout [tap][Cin][Cout]")
    ci, co = x.shape[-1], w.shape[0]
    m = x.numel() // ci
    shape = tuple(x.shape[:-1]) + (co,)
    if out is None:
        out = torch.empty(shape, dtype=torch.float32, device=x.device)
    elif tuple(out.shape) != shape or out.dtype != torch.float32 or not out.is_contiguous() or out.device != x.device:
        raise L.HipExtensionError("`out` must be a contiguous fp32 %s tensor on %s" % (shape, x.device))
    img = _smallk_image(w, owner, ntaps * ci, co)
    plane, d = (x.shape[2] * x.shape[3], x.shape[1]) if ntaps == 3 else (1, 1)
    lib = L.lib()
    def call():
        return L.check(lib.mi_smallk_fwd_f32(L.ptr(x), L.ptr(img), L.ptr(bias), L.ptr(out), int(relu), m, ci, co, ntaps, plane, d,
                                             L.stream()), "mi_smallk_fwd_f32")
    _prof_run("fwd", 2.0 * m * co * ci * ntaps, call)
    return out


# Pre-cut weight images of the SimSiam 2-D encoder's 3 x 3 / stride-1 layers (conv_p2d.hip).  An engine that knows when the weights
# change (SimSiamStepEngine: behind its SGD kernel) keeps them in ACTIVE_P2D = {(weight address, dgrad): image} for the duration of its
# step and re-cuts them all in one or two launches; everywhere else the image lives on the parameter and is rebuilt when the
# parameter's version (torch ops) or the weight epoch (raw-pointer writes) moved.
ACTIVE_P2D = None


def p2d_usable(x_shape, ci, co, k3, stride, p3, dil=None):
    """True when conv_p2d.hip takes this 2-D convolution (forward of x / data gradient onto x): (N, H, W, C) -> C channels, 3 x 3,
    stride 1, padding 1, (W, C) one of the SimSiam 2-D encoder's three at --bbox 36."""
    if len(x_shape) != 4 or ci != co or tuple(k3) != (1, 3, 3) or stride != 1 or tuple(p3) != (0, 1, 1) or dil is not None:
        return False
    if not _arith_bf16x3():
        return False
    n, h, w, _ = x_shape
    return bool(L.lib().mi_conv2d_p2d_usable(int(n), int(h), int(w), int(ci)))


def _stem3_ok(co):
    """Conv2d(1, co, 3, padding=1) on its own kernels (conv_p2d.hip stem3_*): co a multiple of 4 that divides the block evenly."""
    return 4 <= co <= 64 and co % 4 == 0 and 256 % (co // 4) == 0 and not os.environ.get("MI_NO_P2D")


def p2d_prep(items):
    """items: [(weight, dgrad flag, image tensor)] - all cut in one launch per 16 (mi_conv2d_p2d_prep) on the current stream."""
    import ctypes
    n = len(items)
    if not n:
        return
    ws = (ctypes.c_void_p * n)(*[it[0].data_ptr() for it in items])
    imgs = (ctypes.c_void_p * n)(*[it[2].data_ptr() for it in items])
    dg = (ctypes.c_int * n)(*[int(bool(it[1])) for it in items])
    ch = (ctypes.c_int * n)(*[int(it[0].shape[0]) for it in items])
    cast = lambda a: ctypes.cast(a, ctypes.c_void_p)
    L.check(L.lib().mi_conv2d_p2d_prep(cast(ws), cast(imgs), cast(dg), cast(ch), n, L.stream()), "mi_conv2d_p2d_prep")


def _p2d_image(w, dgrad):
    dgrad = int(bool(dgrad))
    if ACTIVE_P2D is not None and PROFILE is None:
        img = ACTIVE_P2D.get((w.data_ptr(), dgrad))
        if img is not None:
            return img
    if not _phys_ok(w):
        raise L.HipExtensionError("conv weight is not in kernel layout [tap][Cin][Cout]")
    key = (w.data_ptr(), w._version, WEIGHT_EPOCH)
    cache = getattr(w, "_mi_p2d", None)
    if cache is None:
        cache = {}
        try:
            w._mi_p2d = cache
        except AttributeError:
            pass
    ent = cache.get(dgrad)
    if ent is None or ent[0] != key:
        img = ent[1] if (ent is not None and ent[1].device == w.device) else torch.empty(int(L.lib().mi_conv2d_p2d_wimg_bytes(int(w.shape[0]))), dtype=torch.uint8,
                                                         device=w.device)
        p2d_prep([(w, dgrad, img)])
        ent = (key, img)
        cache[dgrad] = ent
    return ent[1]


def _p2d_call(a, w, dgrad, res, mask, relu, tag):
    n, h, wd, c = a.shape
    out = torch.empty_like(a)
    img = _p2d_image(w, dgrad)
    lib = L.lib()
    def call():
        return L.check(lib.mi_conv2d_p2d_f32(L.ptr(a), L.ptr(img), L.ptr(out), L.ptr(res), L.ptr(mask), int(relu), n, h, wd, c,
                                             L.stream()), "mi_conv2d_p2d_f32")
    _prof_run(tag, 2.0 * a.numel() * c * 9, call)
    return out


def conv_fwd(x, w, k, stride, pad, res=None, relu=False, dil=None, owner=None, inference=False):
    """y = act(conv(x, w) + res).  x: (N,D,H,W,Ci) or (N,H,W,Ci) channels-last; w in kernel layout.
    dil: per-axis dilation (stride 1 only).  inference: the caller ran inference_mode() where the user's grad mode is visible
    (this function is also called from autograd.Function.forward, where grad mode is always off); `owner`: the tensor that
    outlives a temporary view `w` and keeps its cached weight image."""
    _f32c(x, "x")
    if not _phys_ok(w):
        raise L.HipExtensionError("conv weight is not in kernel layout [tap][Cin][Cout]")
    nd5 = x.dim() == 5
    k3, p3 = _k3(k, nd5), _p3(pad, nd5)
    if res is None:
        kind = _d32_kind(_as5d(x).shape, x.shape[-1], w.shape[0], k3, stride, p3, _k3(dil, nd5) if dil is not None else None, inference)
        if kind == 4:                                 # inference, 1 x 1 (the transposed convolutions' products): tile-resident kernel
            return _d32_call(x, w, None, relu, kind, owner=owner)
        taps = _smallk_taps(x, w, k3, stride, p3, dil, nd5, inference)
        if taps:                                      # inference, short reduction: no tile pipeline (conv_smallk.hip)
            return _smallk_call(x, w, None, relu, taps, owner=owner)
        if kind:                                      # inference, 3 x 3: patch-resident direct kernel (conv_d32.hip)
            return _d32_call(x, w, None, relu, kind, owner=owner)
    if (not nd5 and x.is_cuda and x.shape[-1] == 1 and tuple(k3) == (1, 3, 3) and stride == 1 and tuple(p3) == (0, 1, 1) and dil is None
            and res is None and not relu and _stem3_ok(int(w.shape[0]))):
        n, h, wd, _ = x.shape
        y = torch.empty((n, h, wd, int(w.shape[0])), dtype=torch.float32, device=x.device)
        lib = L.lib()
        def call():
            return L.check(lib.mi_conv2d_stem3_fwd_f32(L.ptr(x), L.ptr(w), L.ptr(y), n, h, wd, int(w.shape[0]), L.stream()),
                           "mi_conv2d_stem3_fwd_f32")
        _prof_run("fwd", 2.0 * y.numel() * 9, call)
        return y
    if not nd5 and x.is_cuda and p2d_usable(x.shape, x.shape[-1], w.shape[0], k3, stride, p3, dil):
        if res is not None:
            _f32c(res, "res")
        return _p2d_call(x, owner if owner is not None else w, False, res, None, relu, "fwd")
    x5 = _as5d(x)
    n, d, h, wd, ci = x5.shape
    co = w.shape[0]
    lib = L.lib()
    if dil is not None and tuple(dil) != (1, 1, 1):
        d3 = _k3(dil, nd5)
        if stride != 1:
            raise L.HipExtensionError("dilated convolution needs stride 1")
        do, ho, wo = (v + 2 * p - dl * (kk - 1) for v, kk, p, dl in zip((d, h, wd), k3, p3, d3))
        y = torch.empty((n, do, ho, wo, co) if nd5 else (n, ho, wo, co), dtype=torch.float32, device=x.device)
        ws = _ws(lib.mi_convnd_dil_workspace_bytes(n, d, h, wd, ci, co, *k3, *p3, *d3), x.device, "conv")
        def call():
            return L.check(lib.mi_convnd_dil_fwd_f32(L.ptr(x), L.ptr(w), L.ptr(y), L.ptr(res), int(relu), n, d, h,
                wd, ci, co, *k3, *p3, *d3, L.ptr(ws), ws.numel(), L.stream()), "mi_convnd_dil_fwd_f32")
        _prof_run("fwd", 2.0 * n * do * ho * wo * co * ci * k3[0] * k3[1] * k3[2], call)
        return y
    do, ho, wo = _out_dims(x5.shape, k3, stride, p3)
    y = torch.empty((n, do, ho, wo, co) if nd5 else (n, ho, wo, co), dtype=torch.float32, device=x.device)
    img = _cached_image(w, False, n, d, h, wd, k3, stride, p3) if nd5 else None
    if img is not None:
        ws = _ws(lib.mi_conv3d_direct_workspace_bytes(n, ci), x.device, "conv")
        L.check(lib.mi_conv3d_direct_f32(L.ptr(x), L.ptr(img), L.ptr(y), L.ptr(res), None, int(relu), n, d, h, wd, ci,
                                         L.ptr(ws), ws.numel(), L.stream()), "mi_conv3d_direct_f32")
        return y
    if nd5 and _cube2_final(lib, n, d, h, wd, ci, co, k3, stride, p3):
        ws = _cube2_ws(lib, n, ci, x.device)
        def call():
            return _cube2_call(lib, x, w, y, res, None, int(relu), 0, n, ci, ws)
        _prof_run("fwd", 2.0 * n * do * ho * wo * co * ci * 27, call)
        return y
    ws = _ws(lib.mi_convnd_workspace_bytes(n, d, h, wd, ci, co, *k3, stride, *p3), x.device, "conv")
    def call():
        return L.check(lib.mi_convnd_fwd_f32(L.ptr(x), L.ptr(w), L.ptr(y), L.ptr(res), int(relu), n, d, h, wd, ci,
            co, *k3, stride, *p3, L.ptr(ws), ws.numel(), L.stream()), "mi_convnd_fwd_f32")
    _prof_run("fwd", 2.0 * n * do * ho * wo * co * ci * k3[0] * k3[1] * k3[2], call)
    return y


def _cube2_final(lib, n, d, h, wd, ci, co, k3, stride, p3):
    """3^3 / stride 1 / padding 1 on a 2 x 2 x 2 volume (layer3, feature_3d): conv_cube2.hip, final in one launch
    (mi_conv3d_cube2_f32; MI_CUBE2_REDUCE=1 keeps the partial sums + reduce launch of mi_convnd_*)."""
    if os.environ.get("MI_CUBE2_REDUCE") or os.environ.get("MI_CONV_ARITH", "")[:1] == "f":
        return False
    return (tuple(k3) == (3, 3, 3) and tuple(p3) == (1, 1, 1) and
            bool(lib.mi_conv3d_cube2_usable(n, d, h, wd, ci, co, 3, stride, 1)))


def _cube2_ws(lib, n, c, device):
    # arrival counters in front of the partial sums: zero at allocation, left zero by every call; one buffer per stream
    return L.workspace(lib.mi_conv3d_cube2_workspace_bytes(n, c), device, "cube2", init=lambda b: b.zero_())


def _cube2_call(lib, a, w, out, res, mask, relu, dgrad, n, c, ws):
    rc = lib.mi_conv3d_cube2_f32(L.ptr(a), L.ptr(w), L.ptr(out), L.ptr(res), L.ptr(mask), relu, dgrad, n, c, L.ptr(ws),
                                 ws.numel(), L.stream())
    if rc:
        L.drop_workspace(a.device, "cube2")         # its counters can no longer be trusted
    return L.check(rc, "mi_conv3d_cube2_f32")


def conv_dgrad(dy, w, in_shape, k, stride, pad, res=None, mask=None, dil=None):
    _f32c(dy, "dy")
    nd5 = len(in_shape) == 5
    k3, p3 = _k3(k, nd5), _p3(pad, nd5)
    shape5 = tuple(in_shape) if nd5 else (in_shape[0], 1) + tuple(in_shape[1:])
    n, d, h, wd, ci = shape5
    co = w.shape[0]
    if not nd5 and dy.is_cuda and p2d_usable(tuple(in_shape), ci, co, k3, stride, p3, dil):
        if res is not None:
            _f32c(res, "res")
        if mask is not None:
            _f32c(mask, "mask")
        return _p2d_call(dy, w, True, res, mask, False, "dgrad")
    dx = torch.empty(tuple(in_shape), dtype=torch.float32, device=dy.device)
    lib = L.lib()
    flops = 2.0 * dy.numel() * ci * k3[0] * k3[1] * k3[2]
    if dil is not None and tuple(_k3(dil, nd5)) != (1, 1, 1):
        d3 = _k3(dil, nd5)
        ws = _ws(lib.mi_convnd_dil_workspace_bytes(n, d, h, wd, ci, co, *k3, *p3, *d3), dy.device, "conv")
        def call():
            return L.check(lib.mi_convnd_dil_dgrad_f32(L.ptr(dy), L.ptr(w), L.ptr(dx), L.ptr(res), L.ptr(mask), n, d,
                h, wd, ci, co, *k3, *p3, *d3, L.ptr(ws), ws.numel(), L.stream()), "mi_convnd_dil_dgrad_f32")
        _prof_run("dgrad", flops, call)
        return dx
    img = _cached_image(w, True, n, d, h, wd, k3, stride, p3) if (nd5 and dil is None) else None
    if img is not None:
        ws = _ws(lib.mi_conv3d_direct_workspace_bytes(n, ci), dy.device, "conv")
        L.check(lib.mi_conv3d_direct_f32(L.ptr(dy), L.ptr(img), L.ptr(dx), L.ptr(res), L.ptr(mask), 0, n, d, h, wd, ci,
                                         L.ptr(ws), ws.numel(), L.stream()), "mi_conv3d_direct_f32")
        return dx
    if nd5 and dil is None and _cube2_final(lib, n, d, h, wd, ci, co, k3, stride, p3):
        ws = _cube2_ws(lib, n, ci, dy.device)
        def call():
            return _cube2_call(lib, dy, w, dx, res, mask, 0, 1, n, ci, ws)
        _prof_run("dgrad", flops, call)
        return dx
    ws = _ws(lib.mi_convnd_workspace_bytes(n, d, h, wd, ci, co, *k3, stride, *p3), dy.device, "conv")
    def call():
        return L.check(lib.mi_convnd_dgrad_f32(L.ptr(dy), L.ptr(w), L.ptr(dx), L.ptr(res), L.ptr(mask), n, d, h, wd,
            ci, co, *k3, stride, *p3, L.ptr(ws), ws.numel(), L.stream()), "mi_convnd_dgrad_f32")
    _prof_run("dgrad", flops, call)
    return dx


def conv_fwd_s2_block(x, w, w_ds):
    """Forward of a BasicBlock's stride-2 front in ONE launch (csrc/conv_s2.hip): (relu(conv(x; w, 3x3x3 stride 2 pad 1)),
    conv(x; w_ds, 1x1 stride 2)).  None when the shape is not one of the encoder's two (the caller runs the generic launches)."""
    if x.dim() != 5 or w_ds is None:
        return None
    n, g, g2, g3, ci = x.shape
    co = w.shape[0]
    lib = L.lib()
    if not x.is_cuda or g != g2 or g != g3 or not lib.mi_conv3d_s2_fwd_usable(n, g, ci, co) or not (_phys_ok(w) and _phys_ok(w_ds)):
        return None
    _f32c(x, "x")
    hmid = torch.empty((n, g // 2, g // 2, g // 2, co), dtype=torch.float32, device=x.device)
    r = torch.empty_like(hmid)
    img = ACTIVE_IMAGES.get_s2(w, False) if (ACTIVE_IMAGES is not None and PROFILE is None) else None
    if img is not None:                                   # the engine keeps the image current: no image build here
        L.check(lib.mi_conv3d_s2_fwd_img_f32(L.ptr(x), L.ptr(img), L.ptr(hmid), L.ptr(r), n, g, ci, co, L.stream()),
                "mi_conv3d_s2_fwd_img_f32")
        return hmid, r
    ws = _ws(lib.mi_conv3d_s2_fwd_workspace_bytes(ci, co), x.device, "conv_s2f")
    def call():
        return L.check(lib.mi_conv3d_s2_fwd_f32(L.ptr(x), L.ptr(w), L.ptr(w_ds), L.ptr(hmid), L.ptr(r), n, g, ci, co, L.ptr(ws),
                                                ws.numel(), L.stream()), "mi_conv3d_s2_fwd_f32")
    # algorithmic FLOPs of both convolutions the launch replaces (27 taps + the 1x1 shortcut)
    _prof_run("fwd", 2.0 * hmid.numel() * ci * 28, call)
    return hmid, r


def conv_dgrad_s2_block(dh, d2, w, w_ds, in_shape, res=None, mask=None):
    """Data gradient of a BasicBlock's stride-2 front in ONE launch (csrc/conv_s2.hip): conv_dgrad(dh; w, 3x3x3 stride 2 pad 1)
    + conv_dgrad(d2; w_ds, 1x1 stride 2) (+ res), times (mask > 0).  Returns None when the shape is not one of the encoder's two
    (the caller then runs the two generic launches)."""
    n, g, _, _, ci = in_shape
    co = w.shape[0]
    lib = L.lib()
    if not dh.is_cuda or not lib.mi_conv3d_s2_dgrad_usable(n, g, ci, co):
        return None
    if in_shape[1] != in_shape[2] or in_shape[2] != in_shape[3] or not (_phys_ok(w) and (w_ds is None or _phys_ok(w_ds))):
        return None
    _f32c(dh, "dh")
    if d2 is not None:
        _f32c(d2, "d2")
    dx = torch.empty(tuple(in_shape), dtype=torch.float32, device=dh.device)
    img = ACTIVE_IMAGES.get_s2(w, True) if (ACTIVE_IMAGES is not None and PROFILE is None and d2 is not None) else None
    if img is not None:                                   # the engine keeps the images current: no image build here
        L.check(lib.mi_conv3d_s2_dgrad_img_f32(L.ptr(dh), L.ptr(d2), L.ptr(img), L.ptr(dx), L.ptr(res), L.ptr(mask), n, g, ci, co,
                                               L.stream()), "mi_conv3d_s2_dgrad_img_f32")
        return dx
    ws = _ws(lib.mi_conv3d_s2_dgrad_workspace_bytes(ci, co), dh.device, "conv_s2")
    def call():
        return L.check(lib.mi_conv3d_s2_dgrad_f32(L.ptr(dh), L.ptr(d2), L.ptr(w), L.ptr(w_ds), L.ptr(dx), L.ptr(res), L.ptr(mask), n, g,
                                                  ci, co, L.ptr(ws), ws.numel(), L.stream()), "mi_conv3d_s2_dgrad_f32")
    # algorithmic FLOPs of both convolutions the launch replaces (27 taps + the 1x1 shortcut)
    _prof_run("dgrad", 2.0 * dh.numel() * ci * (27 + (1 if d2 is not None else 0)), call)
    return dx


# Deferred split-K reduction of the weight gradients: while DEFERRED_WGRADS is a list (MocoStepEngine sets it around the
# backward pass), every wgrad launch leaves its slabs in a slab buffer of its own and registers (slabs, target, splits,
# elements) here; flush_wgrad_reduces() sums them all in one launch (one per gradient bucket under data parallelism)
# instead of one small reduce launch behind each of the 18 weight-gradient kernels of a step.
DEFERRED_WGRADS = None


def flush_wgrad_reduces():
    global DEFERRED_WGRADS
    items = DEFERRED_WGRADS
    if not items:
        return
    import ctypes
    n = len(items)
    slabs = (ctypes.c_void_p * n)(*[it[0].data_ptr() for it in items])
    outs = (ctypes.c_void_p * n)(*[it[1].data_ptr() for it in items])
    cnt = (ctypes.c_int * n)(*[it[2] for it in items])
    elems = (ctypes.c_long * n)(*[it[3] for it in items])
    L.check(L.lib().mi_splitk_reduce_batch(ctypes.cast(slabs, ctypes.c_void_p), ctypes.cast(outs, ctypes.c_void_p),
                                           ctypes.cast(cnt, ctypes.c_void_p), ctypes.cast(elems, ctypes.c_void_p), n,
                                           L.stream()), "mi_splitk_reduce_batch")
    del items[:]


# Weight gradients of the small layers on a side stream: while SIDE_WGRADS is a list (MocoStepEngine sets it around the
# backward pass), the weight-gradient launches of layers with at most SIDE_ROWS_MAX output rows (everything but the stem at
# the encoder's shapes) are not issued where autograd reaches them but collected here; the engine issues a whole stage's worth
# on ONE fork of a side stream when the data-gradient chain leaves the stage: layer3's and the head's next to layer2's
# data gradients, layer2's next to layer1's, layer1's next to the HBM-bound stem chain.  One cross-stream edge per stage - an
# edge per launch costs more than it returns (r03_experiments.txt items 6 and 9).
SIDE_WGRADS = None
SIDE_ROWS_MAX = int(os.environ.get("CETPICK_SIDE_ROWS", "32768"))


def _side_or_now(fn, rows):
    """True: queued for the side stream (MocoStepEngine issues it at the stage boundary); False: launched now."""
    # (bench.py's roofline pass - PROFILE - times every call by itself: closures run where they are; convolution jobs are still
    # collected, so that the pass times the launches the step really makes - run_wgrad_jobs times a group as one call)
    if SIDE_WGRADS is not None and rows <= SIDE_ROWS_MAX and (PROFILE is None or isinstance(fn, _WgradJob)):
        SIDE_WGRADS.append(fn)
        return True
    fn()
    return False


class _WgradJob:
    """One deferred convolution weight gradient (slab form): callable like the closures next to it in SIDE_WGRADS; jobs of one
    geometry that are issued together go out as ONE launch (run_wgrad_jobs -> mi_convnd_wgrad_slabs_batch_f32)."""
    __slots__ = ("x", "dy", "tgt", "slab", "geom", "param", "flops")

    def __init__(self, x, dy, tgt, slab, geom, param=None, flops=0.0):
        self.x, self.dy, self.tgt, self.slab, self.geom, self.param, self.flops = x, dy, tgt, slab, geom, param, flops

    def __call__(self):
        import ctypes
        splits = ctypes.c_int(0)
        L.check(L.lib().mi_convnd_wgrad_slabs_f32(L.ptr(self.x), L.ptr(self.dy), L.ptr(self.tgt), *self.geom, L.ptr(self.slab),
                                                  self.slab.numel(), ctypes.addressof(splits), L.stream()),
                "mi_convnd_wgrad_slabs_f32")
        if splits.value > 1:
            DEFERRED_WGRADS.append((self.slab, self.tgt, int(splits.value), self.tgt.numel()))


WGRAD_BATCH = os.environ.get("CETPICK_WGRAD_BATCH", "1") != "0"
WGRAD_BATCH_MAX = 4

# Outside a MocoStepEngine step (plain autograd: loss.backward() of any trainer) the same grouping happens at the END of the backward
# pass: the jobs wait in _BACKWARD_END and one autograd final callback issues them (run_wgrad_jobs + the slab reduce) on the caller's
# stream.  Engine and plain sequence thus launch the same kernels on the same chains - their gradients stay equal bit for bit
# (tests/test_trainer_gpu.py::test_engine_step_equals_plain_sequence_bitwise) although a batched layer1 gradient is cut into other
# chains than a single one.
_BACKWARD_END = None
_BACKWARD_END_TASK = -1          # autograd graph-task id of the pass the queued jobs belong to


def _defer_to_backward_end(job):
    global _BACKWARD_END, _BACKWARD_END_TASK
    try:
        # (one callback per job: the first one to run flushes everything, the others find the list empty)
        torch.autograd.Variable._execution_engine.queue_callback(_flush_backward_end)
    except RuntimeError:                                   # not inside a backward pass (a direct call): launch now
        return False
    task = torch._C._current_graph_task_id()
    if _BACKWARD_END is not None and task != _BACKWARD_END_TASK:
        #  jobs of a backward pass that RAISED (its final callbacks never ran): their operands belong to a dead pass -
        # dropped, never launched into the gradients of this one
        for stale in _BACKWARD_END:
            if stale.param is not None:
                stale.param._mi_wgrad_pending = False
        _BACKWARD_END = None
    if _BACKWARD_END is None:
        _BACKWARD_END = []
        _BACKWARD_END_TASK = task
    _BACKWARD_END.append(job)
    if job.param is not None:
        job.param._mi_wgrad_pending = True                 # (a second contribution to this parameter flushes first: conv_wgrad_into)
    return True


def _flush_backward_end():
    global _BACKWARD_END, DEFERRED_WGRADS
    items, _BACKWARD_END = _BACKWARD_END, None
    if not items:
        return
    saved, DEFERRED_WGRADS = DEFERRED_WGRADS, []
    try:
        run_wgrad_jobs(items)
        flush_wgrad_reduces()
    finally:
        DEFERRED_WGRADS = saved


def run_wgrad_jobs(items):
    """Issue the collected weight-gradient launches of a stage on the current stream: convolution jobs of one geometry in
    groups of up to four per launch, everything else (and what the library declines) one by one, in collection order."""
    import ctypes
    groups, order = {}, []
    for it in items:
        if isinstance(it, _WgradJob) and it.param is not None:
            it.param._mi_wgrad_pending = False
        if isinstance(it, _WgradJob) and WGRAD_BATCH and DEFERRED_WGRADS is not None:
            key = it.geom + (it.slab.numel(),)
            if key not in groups:
                groups[key] = []
                order.append(groups[key])
            groups[key].append(it)
        else:
            order.append(it)
    lib = L.lib()
    if PROFILE is not None:
        _run_wgrad_jobs_profiled(order)
        return
    for it in order:
        if not isinstance(it, list):
            it()
            continue
        for i0 in range(0, len(it), WGRAD_BATCH_MAX):
            jobs = it[i0:i0 + WGRAD_BATCH_MAX]
            rc = -3
            if len(jobs) > 1:
                n = len(jobs)
                xs = (ctypes.c_void_p * n)(*[j.x.data_ptr() for j in jobs])
                dys = (ctypes.c_void_p * n)(*[j.dy.data_ptr() for j in jobs])
                dws = (ctypes.c_void_p * n)(*[j.tgt.data_ptr() for j in jobs])
                wss = (ctypes.c_void_p * n)(*[j.slab.data_ptr() for j in jobs])
                splits = ctypes.c_int(0)
                rc = lib.mi_convnd_wgrad_slabs_batch_f32(ctypes.cast(xs, ctypes.c_void_p), ctypes.cast(dys, ctypes.c_void_p),
                                                         ctypes.cast(dws, ctypes.c_void_p), ctypes.cast(wss, ctypes.c_void_p), n,
                                                         *jobs[0].geom, jobs[0].slab.numel(), ctypes.addressof(splits), L.stream())
                if rc == 0 and splits.value > 1:
                    for j in jobs:
                        DEFERRED_WGRADS.append((j.slab, j.tgt, int(splits.value), j.tgt.numel()))
            if rc == -3:                                   # MI_E_UNSUPPORTED (or a single job): one launch each
                for j in jobs:
                    j()
            else:
                L.check(rc, "mi_convnd_wgrad_slabs_batch_f32")


def _run_wgrad_jobs_profiled(order):
    """bench.py's roofline pass: a group's launch + the reduce of its slabs is ONE timed call (what the step launches), with the
    group's algorithmic FLOPs; the slabs are reduced here instead of in the stage's deferred reduce."""
    import ctypes
    global DEFERRED_WGRADS
    lib = L.lib()
    for it in order:
        if not isinstance(it, list):
            if isinstance(it, _WgradJob):
                it = [it]
            else:
                it()
                continue
        for i0 in range(0, len(it), WGRAD_BATCH_MAX):
            jobs = it[i0:i0 + WGRAD_BATCH_MAX]

            def call(jobs=jobs):
                global DEFERRED_WGRADS
                saved, DEFERRED_WGRADS = DEFERRED_WGRADS, []
                try:
                    rc = -3
                    if len(jobs) > 1:
                        n = len(jobs)
                        arr = lambda vals: ctypes.cast((ctypes.c_void_p * n)(*vals), ctypes.c_void_p)
                        keep = [arr([j.x.data_ptr() for j in jobs]), arr([j.dy.data_ptr() for j in jobs]),
                                arr([j.tgt.data_ptr() for j in jobs]), arr([j.slab.data_ptr() for j in jobs])]
                        splits = ctypes.c_int(0)
                        rc = lib.mi_convnd_wgrad_slabs_batch_f32(*keep, n, *jobs[0].geom, jobs[0].slab.numel(), ctypes.addressof(splits),
                                                                 L.stream())
                        if rc == 0 and splits.value > 1:
                            for j in jobs:
                                DEFERRED_WGRADS.append((j.slab, j.tgt, int(splits.value), j.tgt.numel()))
                    if rc == -3:
                        for j in jobs:
                            j()
                    else:
                        L.check(rc, "mi_convnd_wgrad_slabs_batch_f32")
                    flush_wgrad_reduces()
                finally:
                    DEFERRED_WGRADS = saved
            _prof_run("wgrad", sum(j.flops for j in jobs), call)


def conv_wgrad_into(x, dy, param, k, stride, pad, dil=None):
    """dW for `param`, written (or accumulated) into param.grad."""
    nd5 = x.dim() == 5
    k3, p3 = _k3(k, nd5), _p3(pad, nd5)
    n, d, h, wd, ci = _as5d(x).shape
    co = dy.shape[-1]
    lib = L.lib()
    if getattr(param, "_mi_wgrad_pending", False):
        # a module applied twice in one backward pass (the symmetric MoCo loss): the first contribution still waits for the end of
        # the pass with this parameter's gradient as its target and its slab buffer - it goes out now, this one accumulates onto it
        if DEFERRED_WGRADS is not None:
            raise L.HipExtensionError("MocoStepEngine: a convolution applied twice in one backward pass (its first weight gradient is "
                                      "still queued for the side stream) - run this model without the step engine")
        _flush_backward_end()
    g, acc = _grad_target(param)
    tgt = torch.empty_like(g) if acc else g
    flops = 2.0 * dy.numel() * ci * k3[0] * k3[1] * k3[2]
    if (not nd5 and x.is_cuda and ci == 1 and tuple(k3) == (1, 3, 3) and stride == 1 and tuple(p3) == (0, 1, 1) and dil is None
            and _stem3_ok(int(co)) and _phys_ok(param)):
        _f32c(x, "x"), _f32c(dy, "dy")
        ws = _ws(lib.mi_conv2d_stem3_workspace_bytes(int(co)), x.device, "stem3")
        def call():
            return L.check(lib.mi_conv2d_stem3_wgrad_f32(L.ptr(x), L.ptr(dy), L.ptr(tgt), n, h, wd, int(co), L.ptr(ws), ws.numel(),
                                                         L.stream()), "mi_conv2d_stem3_wgrad_f32")
        _prof_run("wgrad", flops, call)
        if acc:
            g.add_(tgt)
        return
    # (under bench.py's roofline pass only the jobs the engine will queue take the job form - run_wgrad_jobs times them group by group -,
    # everything else is timed right here, call by call)
    p2d_ws = (int(lib.mi_conv2d_p2d_wgrad_workspace_bytes(n, h, wd, ci))
              if (not nd5 and x.is_cuda and p2d_usable(tuple(x.shape), ci, int(co), k3, stride, p3, dil) and _phys_ok(param)
                  and not os.environ.get("MI_NO_P2D_WGRAD")) else 0)
    if p2d_ws:
        # the 2-D encoder's 3 x 3 / stride-1 layers: voxel-major operands through the transposing LDS read (conv_p2d.hip p2d_wgrad_kernel)
        _f32c(x, "x"), _f32c(dy, "dy")
        ws = _ws(p2d_ws, x.device, "p2d_wgrad")
        def call():
            return L.check(lib.mi_conv2d_p2d_wgrad_f32(L.ptr(x), L.ptr(dy), L.ptr(tgt), n, h, wd, ci, L.ptr(ws), ws.numel(), L.stream()),
                           "mi_conv2d_p2d_wgrad_f32")
        _prof_run("wgrad", flops, call)
        if acc:
            g.add_(tgt)
        return
    slab_form = (not acc and (dil is None or tuple(_k3(dil, nd5)) == (1, 1, 1)) and x.is_cuda and
                 (PROFILE is None or (DEFERRED_WGRADS is not None and SIDE_WGRADS is not None and dy.numel() // co <= SIDE_ROWS_MAX)))
    if slab_form and (DEFERRED_WGRADS is not None or (WGRAD_BATCH and lib.mi_conv3d_direct_usable(n, d, h, wd, ci, co, k3[0], stride, p3[0]) in (1, 2)
                                                      and k3[0] == k3[1] == k3[2] and p3[0] == p3[1] == p3[2])):
        nbytes = lib.mi_convnd_workspace_bytes(n, d, h, wd, ci, co, *k3, stride, *p3)
        slab = getattr(param, "_mi_slabs", None)
        if slab is None or slab.numel() < nbytes or slab.device != x.device:
            slab = torch.empty(int(nbytes), dtype=torch.uint8, device=x.device)     # lives with the parameter
            if not getattr(param, "_mi_slabs_pinned", False):
                param._mi_slabs = slab
            # (pinned: a captured hipGraph writes and reads the old buffer on every replay - MocoStepEngine pins the slabs
            # when it captures; an eager call that needs more space gets a buffer of its own and the graph's stays alive)
        _f32c(x, "x"), _f32c(dy, "dy")
        job = _WgradJob(x, dy, tgt, slab, (n, d, h, wd, ci, co) + tuple(k3) + (stride,) + tuple(p3), param, flops)
        if DEFERRED_WGRADS is not None:
            if _side_or_now(job, dy.numel() // co):
                param._mi_wgrad_pending = True             # queued: cleared by run_wgrad_jobs (a second contribution before that raises)
            return
        if _defer_to_backward_end(job):                    # plain autograd: layer1's / layer2's weight gradients, batched at the end
            return
    if dil is not None and tuple(_k3(dil, nd5)) != (1, 1, 1):
        d3 = _k3(dil, nd5)
        ws = _ws(lib.mi_convnd_dil_workspace_bytes(n, d, h, wd, ci, co, *k3, *p3, *d3), x.device, "conv")
        def call():
            return L.check(lib.mi_convnd_dil_wgrad_f32(L.ptr(x), L.ptr(dy), L.ptr(tgt), n, d, h, wd, ci, co, *k3,
                *p3, *d3, L.ptr(ws), ws.numel(), L.stream()), "mi_convnd_dil_wgrad_f32")
        _prof_run("wgrad", flops, call)
    else:
        ws = _ws(lib.mi_convnd_workspace_bytes(n, d, h, wd, ci, co, *k3, stride, *p3), x.device, "conv")
        def call():
            return L.check(lib.mi_convnd_wgrad_f32(L.ptr(x), L.ptr(dy), L.ptr(tgt), n, d, h, wd, ci, co, *k3, stride,
                *p3, L.ptr(ws), ws.numel(), L.stream()), "mi_convnd_wgrad_f32")
        _prof_run("wgrad", flops, call)
    if acc:
        g.add_(tgt)


def relu_mask(dy, y, add=None):
    out = torch.empty_like(dy)
    L.check(L.lib().mi_relu_mask(L.ptr(dy), L.ptr(y), L.ptr(add), L.ptr(out), dy.numel(), L.stream()),
            "mi_relu_mask")
    return out


def _stem_fwd_with_stats(x, w, mod):
    """The stem convolution with the batch statistics of the BatchNorm behind it from its epilogue (mi_conv3d_stem_stats_f32:
    no statistics pass over the 67 MB output).  Leaves them in mod.bn_sums for bn_relu_maxpool3d(..., sums=...); None where
    the stem kernel does not apply (the caller then takes the plain convolution and the BatchNorm its own statistics)."""
    mod.bn_sums = None
    if PROFILE is not None or x.dim() != 5 or x.shape[-1] != 1 or (mod.k, mod.stride, mod.pad) != (7, 2, 3) or not _phys_ok(w):
        return None
    lib = L.lib()
    n, d, h, wd, _ = x.shape
    co = w.shape[0]
    nbytes = lib.mi_conv3d_stem_stats_workspace_bytes(n, d, h, wd, co)
    if nbytes == 0:
        return None
    do, ho, wo = [(v + 6 - 7) // 2 + 1 for v in (d, h, wd)]
    y = torch.empty((n, do, ho, wo, co), dtype=torch.float32, device=x.device)
    sums = torch.empty(2 * co, dtype=torch.float64, device=x.device)
    ws = _ws(nbytes, x.device, "stem_stats")
    rc = lib.mi_conv3d_stem_stats_f32(L.ptr(x), L.ptr(w), L.ptr(y), n, d, h, wd, co, L.ptr(sums), L.ptr(ws), ws.numel(),
                                      L.stream())
    if rc == -3:
        return None
    L.check(rc, "mi_conv3d_stem_stats_f32")
    mod.bn_sums = sums
    return y


class GradSlot:
    """A gradient contribution handed from one autograd node to another of the same block instead of through autograd's sum: the
    node behind a residual connection (BatchNorm + residual + ReLU) leaves the residual branch's gradient here, the block's first
    convolution - whose input IS that residual - adds it in its data gradient's epilogue (`res`).  One element-wise launch less
    per identity block and pass; the sum is the same single rounding."""
    __slots__ = ("tensor",)

    def __init__(self):
        self.tensor = None


FUSE_RES_GRAD = os.environ.get("CETPICK_FUSE_RES_GRAD", "1") != "0"


def grad_slot_for(x):
    """A GradSlot when the gradient of `x` will be computed (and the fusion is on), else None."""
    return GradSlot() if (FUSE_RES_GRAD and torch.is_grad_enabled() and x.requires_grad and x.is_cuda and PROFILE is None) else None


class _ConvFn(torch.autograd.Function):
    """y = act(conv(x, W)); W's gradient goes to mod.weight.grad directly."""

    @staticmethod
    def forward(ctx, x, w, mod, relu, mask_dx=False, inference=False, grad_slot=None, dx_slot=None):
        ctx.mod, ctx.relu, ctx.mask_dx = mod, relu, mask_dx
        ctx.grad_slot = grad_slot                        # a gradient of x that arrives here instead of through autograd's sum: added in the epilogue
        ctx.dx_slot = dx_slot                            # ... and this node's own gradient of x LEAVES through a slot (a shortcut convolution:
                                                         # the block's first convolution, which runs later in the backward pass, adds it)
        y = _stem_fwd_with_stats(x, w, mod) if (getattr(mod, "stats_for_bn", False) and not relu) else None
        if y is None:
            y = conv_fwd(x, w, mod.k, mod.stride, mod.pad, None, relu, dil=getattr(mod, "dil", None), owner=mod.weight,
                         inference=inference)
        ctx.save_for_backward(x, y if relu else None)
        ctx.x_needs_grad = x.requires_grad
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y = ctx.saved_tensors
        mod = ctx.mod
        dy = dy.contiguous()
        if ctx.relu:
            dy = relu_mask(dy, y)
        dil = getattr(mod, "dil", None)
        if mod.weight.requires_grad:
            conv_wgrad_into(x, dy, mod.weight, mod.k, mod.stride, mod.pad, dil=dil)
        dx = None
        extra = None
        if ctx.grad_slot is not None:
            extra, ctx.grad_slot.tensor = ctx.grad_slot.tensor, None
        if ctx.x_needs_grad:
            # mask_dx: x is the output of a ReLU whose derivative the producer left to this layer (see basic_block)
            # extra: the gradient x receives through the block's residual connection (GradSlot), added in the epilogue
            if extra is not None and ctx.mask_dx:
                raise L.HipExtensionError("a residual gradient slot and a masked data gradient do not combine")
            dx = conv_dgrad(dy, mod.weight, x.shape, mod.k, mod.stride, mod.pad, extra, x if ctx.mask_dx else None, dil=dil)
            if ctx.dx_slot is not None:
                if ctx.dx_slot.tensor is not None:
                    raise L.HipExtensionError("gradient slot already holds a tensor")
                ctx.dx_slot.tensor, dx = dx, None
        return dx, None, None, None, None, None, None, None


def conv_bias_fwd(x, w, bias, k, stride, pad, relu=False, out=None, pool=False):
    """y = act(conv(x, w) + bias[co]) in one launch (mi_convnd_fwd_bias_f32), inference only: no autograd node.  `out`: a
    contiguous (N, [D,] Ho, Wo, Co) tensor (or a leading-axis slice of one) that receives the result."""
    _f32c(x, "x")
    if not _phys_ok(w):
        raise L.HipExtensionError("conv weight is not in kernel layout [tap][Cin][Cout]")
    nd5 = x.dim() == 5
    k3, p3 = _k3(k, nd5), _p3(pad, nd5)
    if torch.is_grad_enabled() and (x.requires_grad or w.requires_grad):
        raise L.HipExtensionError("conv_bias_fwd is inference only (no autograd node)")
    kind = _d32_kind(_as5d(x).shape, x.shape[-1], w.shape[0], k3, stride, p3, None, True)
    if pool:
        # (y, maxpool2x2(y)) in one launch where the patch-resident 2-D kernel runs and nothing else asks for the result elsewhere;
        # None: the caller pools by itself
        if kind in (1, 3) and POOL_FUSED and PROFILE is None and x.dim() == 4:
            return _d32_call(x, w, _f32c(bias, "bias"), relu, kind, pool=True, out=out)
        return None
    if kind == 4:                                     # 1 x 1: the tile-resident kernel (in front of the short-reduction one)
        return _d32_call(x, w, _f32c(bias, "bias"), relu, kind, out=out)
    taps = _smallk_taps(x, w, k3, stride, p3, None, nd5, True)
    if taps:
        return _smallk_call(x, w, _f32c(bias, "bias"), relu, taps, out=out)
    if kind:
        return _d32_call(x, w, _f32c(bias, "bias"), relu, kind, out=out)
    x5 = _as5d(x)
    n, d, h, wd, ci = x5.shape
    co = w.shape[0]
    do, ho, wo = _out_dims(x5.shape, k3, stride, p3)
    shape = (n, do, ho, wo, co) if nd5 else (n, ho, wo, co)
    if out is None:
        out = torch.empty(shape, dtype=torch.float32, device=x.device)
    elif tuple(out.shape) != shape or out.dtype != torch.float32 or not out.is_contiguous() or out.device != x.device:
        raise L.HipExtensionError("conv_bias_fwd: `out` must be a contiguous fp32 %s tensor on %s" % (shape, x.device))
    lib = L.lib()
    if (ci == 1 and co == 16 and d == 1 and tuple(k3) == (1, 7, 7) and stride == 2 and tuple(p3) == (0, 3, 3) and n <= 65535
            and not os.environ.get("MI_NO_STEM2D")):
        # the detector's first layer: one input channel - a direct kernel instead of 49 single-float gathers per row
        def call():
            return L.check(lib.mi_stem2d_fwd_bias_f32(L.ptr(x), L.ptr(w), L.ptr(_f32c(bias, "bias")), L.ptr(out), int(relu), n, h, wd,
                                                      L.stream()), "mi_stem2d_fwd_bias_f32")
        _prof_run("fwd", 2.0 * n * do * ho * wo * co * 49, call)
        return out
    ws = _ws(lib.mi_convnd_workspace_bytes(n, d, h, wd, ci, co, *k3, stride, *p3), x.device, "conv")
    def call():
        return L.check(lib.mi_convnd_fwd_bias_f32(L.ptr(x), L.ptr(w), L.ptr(out), L.ptr(_f32c(bias, "bias")), int(relu), n, d, h,
            wd, ci, co, *k3, stride, *p3, L.ptr(ws), ws.numel(), L.stream()), "mi_convnd_fwd_bias_f32")
    _prof_run("fwd", 2.0 * n * do * ho * wo * co * ci * k3[0] * k3[1] * k3[2], call)
    return out


CONCAT_DIRECT = os.environ.get("CETPICK_CONCAT_DIRECT", "1") != "0"


def skip_into_concat_ok(conv, bn, x, co_up, up=None, up_bn=None):
    """True when a down-convolution block's last layer (conv -> bn -> ReLU -> pool on x) can write its un-pooled output - the skip
    connection - straight into the concatenation buffer of the up-convolution block that consumes it, AND that block's transposed
    convolution will write the other channels there with its fused epilogue (upconv_bn_relu_concat): inference, both on conv_d32.hip."""
    if not (CONCAT_DIRECT and POOL_FUSED and UPCONV_FUSED and FOLD_EVAL_BN and PROFILE is None and x.is_cuda and x.dim() == 4
            and not torch.is_grad_enabled() and not bn.training and bn.track_running_stats and _arith_bf16x3()):
        return False
    n, h, w, ci = x.shape
    co = conv.co
    if co != co_up or co not in (32, 64) or h % 16 or w % 32:          # the up block: 2 co -> co on (h / 2, w / 2), rows % 8, columns % 16
        return False
    # the concatenation buffer is co_up + co channels wide: both kernels that write into it address it with 32-bit byte offsets and
    # answer MI_E_UNSUPPORTED from 0x7fff0000 bytes on (a 32-slice chunk of a 1024 x 1024 tomogram: 2^31) - the dense form takes over
    if 4 * n * h * w * (co_up + co) >= 0x7fff0000:
        return False
    # ... and the up block's side of the bargain: evaluation-mode BatchNorm that can be folded, weights in kernel layout
    if up_bn is not None and (up_bn.training or not up_bn.track_running_stats):
        return False
    if up is not None and not _phys_ok(up.gemm_view()):
        return False
    k3, p3 = _k3(conv.k, False), _p3(conv.pad, False)
    return _d32_kind((n, 1, h, w, ci), ci, co, k3, conv.stride, p3, None, True) in (1, 3)


def conv_bn(conv, bn, x, relu=False, pool=False, out=None):
    """bn(conv(x), relu) for a HipConv2d / HipConvNd followed by a HipBatchNorm.  At inference (evaluation-mode BatchNorm with
    running statistics, no gradient) the BatchNorm folds into the convolution: w' = w * gamma / sqrt(var + eps) per output
    channel, bias = beta - mean * gamma / sqrt(var + eps), one launch with a bias + ReLU epilogue and no BatchNorm pass over
    the activation (reference: conv -> BatchNorm2d -> ReLU, models/networks/unet.py:198-249,319-399).  The folded weights are
    kept on the BatchNorm module and rebuilt when any of the five tensors they come from changes.  pool: returns (y, maxpool 2 x 2
    ceil-mode of y) - unet.py:198-249's down-convolution block."""
    folded = (not bn.training and bn.track_running_stats and not torch.is_grad_enabled() and x.is_cuda and
              getattr(conv, "dil", None) in (None, (1, 1, 1)) and FOLD_EVAL_BN)
    if not folded:
        if out is not None:
            raise L.HipExtensionError("conv_bn: `out` needs the folded inference path (skip_into_concat_ok)")
        y = bn(conv(x), relu=relu)
        return (y, maxpool2d_ceil(y, 2)) if pool else y
    src = (conv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var)
    key = tuple((t.data_ptr(), t._version) if t is not None else None for t in src) + (float(bn.eps), WEIGHT_EPOCH)
    cache = getattr(bn, "_folded", None)
    if cache is None or cache[0] != key:
        with torch.no_grad():
            inv = torch.rsqrt(bn.running_var.double() + bn.eps)
            sc = inv if bn.weight is None else bn.weight.double() * inv
            sh = -bn.running_mean.double() * sc
            if bn.bias is not None:
                sh = sh + bn.bias.double()
            wf = (conv.weight.double() * sc.view(-1, *([1] * (conv.weight.dim() - 1)))).float()
            cache = (key, wf, sh.float().contiguous())
        if not _phys_ok(wf):
            raise L.HipExtensionError("folded convolution weight left the kernel layout")
        bn._folded = cache
    if pool:
        # pool = True: (y, MaxPool2d(2, ceil_mode)(y)) - in one launch where the patch-resident kernel takes the layer (the pool is
        # its epilogue's by-product), else the pooling pass behind it
        both = conv_bias_fwd(x, cache[1], cache[2], conv.k, conv.stride, conv.pad, relu, pool=True, out=out)
        if both is not None:
            return both
        if out is not None:
            raise L.HipExtensionError("conv_bn: `out` (a channel slice) needs the patch-resident kernel (skip_into_concat_ok)")
        y = conv_bias_fwd(x, cache[1], cache[2], conv.k, conv.stride, conv.pad, relu)
        return y, maxpool2d_ceil(y, 2)
    if out is not None:
        raise L.HipExtensionError("conv_bn: `out` is the pooled form's (pool=True)")
    return conv_bias_fwd(x, cache[1], cache[2], conv.k, conv.stride, conv.pad, relu)


def upconv_bn_relu_concat(up, bn, dec, enc, cat=None):
    """cat(relu(bn(up(dec))), enc) over the channel axis (unet.py:319-399).  At inference one 1 x 1 product + ONE pass that shuffles,
    applies the folded BatchNorm (the transposed convolution's bias included), the ReLU and writes the concatenation
    (mi_upconv_tail_fwd) - instead of shuffle, BatchNorm and concat passes over the feature map."""
    ho, wo = enc.shape[1], enc.shape[2]
    folded = (not bn.training and bn.track_running_stats and not torch.is_grad_enabled() and dec.is_cuda and FOLD_EVAL_BN)
    if not folded:
        if cat is not None:
            raise L.HipExtensionError("upconv_bn_relu_concat: `cat` needs the folded inference path")
        return concat_channels(bn(up(dec, ho, wo), relu=True), enc)
    src = (up.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var)
    key = tuple((t.data_ptr(), t._version) if t is not None else None for t in src) + (float(bn.eps), WEIGHT_EPOCH)
    cache = getattr(bn, "_folded_up", None)
    if cache is None or cache[0] != key:
        with torch.no_grad():
            inv = torch.rsqrt(bn.running_var.double() + bn.eps)
            sc = inv if bn.weight is None else bn.weight.double() * inv
            sh = -bn.running_mean.double() * sc
            if up.bias is not None:
                sh = sh + up.bias.double() * sc
            if bn.bias is not None:
                sh = sh + bn.bias.double()
            cache = (key, sc.float().contiguous(), sh.float().contiguous())
        bn._folded_up = cache
    _f32c(dec, "dec")
    if cat is None:
        _f32c(enc, "enc")                                # (with `cat`, enc is a channel slice of it: checked below)
    n, h, w, _ = dec.shape
    co, ce = up.co, enc.shape[-1]
    lib = L.lib()
    ci = dec.shape[-1]
    if cat is not None:
        # `enc` IS cat[..., co:] (the down-convolution block wrote it there: skip_into_concat_ok): only the first co channels are missing
        if (tuple(cat.shape) != (n, ho, wo, co + ce) or not cat.is_contiguous() or enc.data_ptr() != cat.data_ptr() + 4 * co):
            raise L.HipExtensionError("upconv_bn_relu_concat: `cat` is not the buffer `enc` is a channel slice of")
        wv = up.gemm_view()
        key = (wv.data_ptr(), up.weight._version, WEIGHT_EPOCH, ci, co, "up")
        wc = getattr(up.weight, "_mi_d32_up", None)
        if wc is None or wc[0] != key:
            img = torch.empty((4 * co // 64) * int(lib.mi_conv_d64_image_bytes(ci, 1)), dtype=torch.uint8, device=dec.device)
            L.check(lib.mi_conv_d64_prep_co(L.ptr(wv), L.ptr(img), ci, 4 * co, 1, L.stream()), "mi_conv_d64_prep_co")
            wc = (key, img)
            up.weight._mi_d32_up = wc
        L.check(lib.mi_conv_d32_upconv_fwd_f32(L.ptr(dec), L.ptr(wc[1]), L.ptr(cache[1]), L.ptr(cache[2]), L.ptr(cat), n, h, w, ci, co,
                                               ho, wo, co + ce, L.stream()), "mi_conv_d32_upconv_fwd_f32")
        return cat
    out = torch.empty((n, ho, wo, co + ce), dtype=torch.float32, device=dec.device)
    if (UPCONV_FUSED and _arith_bf16x3() and PROFILE is None and ci in (32, 64, 128) and co % 32 == 0 and 64 <= 4 * co <= 512
            and h % 8 == 0 and w % 16 == 0 and _phys_ok(up.gemm_view())):
        # the product, the pixel shuffle, the folded BatchNorm and the ReLU in ONE launch (conv_d32.hip, 1 x 1 form with the
        # up-convolution epilogue) straight into the concatenation; the encoder feature goes into the other channels
        wv = up.gemm_view()
        key = (wv.data_ptr(), up.weight._version, WEIGHT_EPOCH, ci, co, "up")
        wc = getattr(up.weight, "_mi_d32_up", None)
        if wc is None or wc[0] != key:
            img = torch.empty((4 * co // 64) * int(lib.mi_conv_d64_image_bytes(ci, 1)), dtype=torch.uint8, device=dec.device)
            L.check(lib.mi_conv_d64_prep_co(L.ptr(wv), L.ptr(img), ci, 4 * co, 1, L.stream()), "mi_conv_d64_prep_co")
            wc = (key, img)
            up.weight._mi_d32_up = wc
        rc = lib.mi_conv_d32_upconv_fwd_f32(L.ptr(dec), L.ptr(wc[1]), L.ptr(cache[1]), L.ptr(cache[2]), L.ptr(out), n, h, w, ci, co,
                                            ho, wo, co + ce, L.stream())
        if rc == 0:
            L.check(lib.mi_copy_channels_into(L.ptr(enc), ce, L.ptr(out), co + ce, co, n * ho * wo, L.stream()), "mi_copy_channels_into")
            return out
        if rc != -3:
            L.check(rc, "mi_conv_d32_upconv_fwd_f32")
    t = conv_fwd(dec, up.gemm_view(), 1, 1, 0, owner=up.weight, inference=True)      # (`folded` above: no gradient)
    L.check(L.lib().mi_upconv_tail_fwd(L.ptr(t), L.ptr(cache[1]), L.ptr(cache[2]), L.ptr(enc), L.ptr(out), n, h, w, co, ce, ho, wo,
                                       L.stream()), "mi_upconv_tail_fwd")
    return out


FOLD_EVAL_BN = os.environ.get("CETPICK_FOLD_BN", "1") != "0"
UPCONV_FUSED = os.environ.get("CETPICK_UPCONV_FUSED", "1") != "0"       # the transposed convolution's product with the tail in its epilogue


def convnd_weight_param(co, ci, k3, device=None):
    """Parameter with logical shape (co, ci, kd, kh, kw) over physical storage [kd,kh,kw,ci,co]."""
    phys = torch.empty(*k3, ci, co, device=device)
    return nn.Parameter(phys.permute(4, 3, 0, 1, 2))


class HipConvNd(nn.Module):
    """nn.Conv3d(ci, co, kernel (kd,kh,kw), stride 1, padding (pd,ph,pw), dilation, bias=False), channels-last."""

    def __init__(self, ci, co, k3, pad3, dil3=(1, 1, 1)):
        super().__init__()
        self.ci, self.co, self.k, self.stride, self.pad, self.dil = ci, co, tuple(k3), 1, tuple(pad3), tuple(dil3)
        self.weight = convnd_weight_param(co, ci, k3)
        with torch.no_grad():
            bound = 1.0 / (ci * k3[0] * k3[1] * k3[2]) ** 0.5
            self.weight.uniform_(-bound, bound)

    def forward(self, x, relu=False):
        return _ConvFn.apply(x, self.weight, self, relu, False, inference_mode())


class HipConv3d(nn.Module):
    """nn.Conv3d(ci, co, k, stride, padding, bias=False) on channels-last activations."""

    def __init__(self, ci, co, k, stride=1, pad=0):
        super().__init__()
        self.ci, self.co, self.k, self.stride, self.pad = ci, co, k, stride, pad
        self.weight = conv_weight_param(co, ci, k)
        with torch.no_grad():
            # nn.Conv3d default init: kaiming_uniform(a=sqrt(5)) == U(-1/sqrt(fan_in), 1/sqrt(fan_in))
            bound = 1.0 / (ci * k ** 3) ** 0.5
            self.weight.uniform_(-bound, bound)

    def forward(self, x, relu=False, mask_dx=False):
        return _ConvFn.apply(x, self.weight, self, relu, mask_dx, inference_mode())


class HipConv2d(nn.Module):
    """nn.Conv2d(ci, co, k, stride, padding, bias=False) on channels-last (N,H,W,C) activations."""

    def __init__(self, ci, co, k, stride=1, pad=0):
        super().__init__()
        self.ci, self.co, self.k, self.stride, self.pad = ci, co, k, stride, pad
        self.weight = conv2d_weight_param(co, ci, k)
        with torch.no_grad():
            bound = 1.0 / (ci * k * k) ** 0.5
            self.weight.uniform_(-bound, bound)

    def forward(self, x, relu=False, grad_slot=None, dx_slot=None):
        return _ConvFn.apply(x, self.weight, self, relu, False, inference_mode(), grad_slot, dx_slot)


class _LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, mod, inference=False):
        m, ci = x.shape
        w5 = w.view(w.shape[0], ci, 1, 1, 1) if w.dim() == 5 else _as5(w)
        y = None
        fuse = getattr(mod, "_fuse_bn", None)
        if fuse is not None:
            mod._bn_pre = None
        if fuse is not None and fuse[2] and PROFILE is None and _phys_ok(w5):
            # SyncBN across ranks (linear_bn): this rank's column sums come out of the product's epilogue and wait in
            # mod._bn_pre for the BatchNorm node, which all-reduces them and applies
            _f32c(x, "x")
            lib = L.lib()
            co = w5.shape[0]
            y = torch.empty((m, co), dtype=torch.float32, device=x.device)
            sums = torch.empty(2 * co, dtype=torch.float64, device=x.device)
            rc = lib.mi_linear_stats_fwd_f32(L.ptr(x), L.ptr(w5), L.ptr(b), L.ptr(y), L.ptr(sums), m, ci, co, L.stream())
            if rc == -3:
                y = None
            else:
                L.check(rc, "mi_linear_stats_fwd_f32")
                mod._bn_pre = ("sums", sums)
        elif fuse is not None and PROFILE is None and _phys_ok(w5):
            # Linear + BatchNorm1d (+ ReLU) in one launch (linear_bn): the BatchNorm's output and saved statistics wait in
            # mod._bn_pre for the BatchNorm node; this node's own output is the Linear's, as always
            bn, bn_relu, _ = fuse
            _f32c(x, "x")
            lib = L.lib()
            co = w5.shape[0]
            y = torch.empty((m, co), dtype=torch.float32, device=x.device)
            yb = torch.empty((m, co), dtype=torch.float32, device=x.device)
            save = torch.empty(2 * co, dtype=torch.float32, device=x.device)
            track = bn.track_running_stats and bn.training
            rc = lib.mi_linear_bn_fwd_f32(L.ptr(x), L.ptr(w5), L.ptr(b), L.ptr(y), L.ptr(yb), m, ci, co, L.ptr(bn.weight),
                                          L.ptr(bn.bias), bn.eps, bn.momentum, L.ptr(bn.running_mean if track else None),
                                          L.ptr(bn.running_var if track else None),
                                          L.ptr(bn.num_batches_tracked if track else None), L.ptr(save), int(bn_relu),
                                          L.stream())
            if rc == -3:
                y = None                                  # declined (rows, arithmetic switch): the two launches
            else:
                L.check(rc, "mi_linear_bn_fwd_f32")
                mod._bn_pre = (yb, save)
        if y is not None:
            pass
        elif b is not None and PROFILE is None and _phys_ok(w5):
            # the bias rides in the epilogue of the GEMM launch (or of its split-K reduce)
            _f32c(x, "x")
            lib = L.lib()
            co = w5.shape[0]
            y = torch.empty((m, co), dtype=torch.float32, device=x.device)
            ws = _ws(lib.mi_conv3d_workspace_bytes(m, 1, 1, 1, ci, co, 1, 1, 0), x.device, "conv")
            L.check(lib.mi_linear_fwd_f32(L.ptr(x), L.ptr(w5), L.ptr(b), L.ptr(y), m, ci, co, L.ptr(ws), ws.numel(),
                                          L.stream()), "mi_linear_fwd_f32")
        else:
            y = conv_fwd(x.view(m, 1, 1, 1, ci), w5, 1, 1, 0, owner=w, inference=inference).view(m, -1)
            if b is not None:
                L.check(L.lib().mi_bias_add(L.ptr(y), L.ptr(b), m, y.shape[1], L.stream()), "mi_bias_add")
        ctx.mod = mod
        ctx.save_for_backward(x)
        ctx.x_needs_grad = x.requires_grad
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        mod = ctx.mod
        dy = dy.contiguous()
        m, ci = x.shape
        co = dy.shape[1]
        lib = L.lib()
        if mod.weight.requires_grad:
            g, acc = _grad_target(mod.weight)
            tgt = torch.empty_like(g) if acc else g
            def launch(g=g, acc=acc, tgt=tgt):          # (bound now: the names are reused for the bias below)
                ws = _ws(lib.mi_conv3d_workspace_bytes(m, 1, 1, 1, ci, co, 1, 1, 0), x.device, "conv")
                L.check(lib.mi_conv3d_wgrad_f32(L.ptr(x), L.ptr(dy), L.ptr(tgt), m, 1, 1, 1, ci, co, 1, 1, 0,
                                                L.ptr(ws), ws.numel(), L.stream()), "mi_conv3d_wgrad_f32")
                if acc:
                    g.add_(tgt)
            _side_or_now(launch, m)
        if mod.bias is not None and mod.bias.requires_grad:
            g, acc = _grad_target(mod.bias)
            tgt = torch.empty_like(g) if acc else g
            def launch_b(g=g, acc=acc, tgt=tgt):       # the bias gradient is a parameter gradient too: same side-stream batch
                ws = _ws(lib.mi_colreduce_workspace_bytes(m, co), x.device, "colreduce")
                sums = torch.empty(2 * co, dtype=torch.float64, device=x.device)
                L.check(lib.mi_colsum(L.ptr(dy), m, co, L.ptr(tgt), L.ptr(sums), L.ptr(ws), ws.numel(), L.stream()),
                        "mi_colsum")
                if acc:
                    g.add_(tgt)
            _side_or_now(launch_b, m)
        dx = None
        if ctx.x_needs_grad:
            dx = conv_dgrad(dy.view(m, 1, 1, 1, co), _as5(mod.weight), (m, 1, 1, 1, ci), 1, 1, 0).view(m, ci)
        return dx, None, None, None, None


def _as5(w2):
    """(out, in) weight with strides (1, out) viewed as (out, in, 1, 1, 1) in kernel layout."""
    out_f, in_f = w2.shape
    return torch.as_strided(w2, (out_f, in_f, 1, 1, 1), (1, out_f, in_f * out_f, in_f * out_f, in_f * out_f))


class HipLinear(nn.Module):
    def __init__(self, in_f, out_f, bias=True):
        super().__init__()
        self.in_f, self.out_f = in_f, out_f
        self.weight = linear_weight_param(out_f, in_f)
        self.bias = nn.Parameter(torch.empty(out_f)) if bias else None
        with torch.no_grad():
            bound = 1.0 / in_f ** 0.5
            self.weight.uniform_(-bound, bound)
            if bias:
                self.bias.uniform_(-bound, bound)

    def forward(self, x):
        if not _phys_ok(self.weight):
            raise L.HipExtensionError("linear weight is not in kernel layout [in][out]")
        return _LinearFn.apply(_f32c(x, "x"), self.weight, self.bias, self, inference_mode())


# ------------------------------------------------------------------------------------------------
# batch norm (+ReLU), SyncBN-capable
# ------------------------------------------------------------------------------------------------
def _dist_world():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size()
    return 1


FORCE_COLLECTIVES = False     # tests: issue every collective on a 1-rank group too (one-GPU rehearsal of the N>1 path)


# Collectives under hipGraph capture are issued SYNCHRONOUSLY (async_op=False), always.  A collective issued with
# async_op=True while its stream is being captured ends up polled by the process group's watchdog thread, whose
# hipEventQuery on the Work's end event - last recorded in a capturing stream - raises hipErrorCapturedEvent and
# terminates the process ("operation not permitted on an event last recorded in a capturing stream"; measured with
# tools/diag_captured_event.py / tools/diag_teardown.py, profiles/r02_teardown_diag.txt: 4 of 4 captured runs die with async
# works, 0 of 4 without).  That was the intermittent abort of
# the N > 1 captured step that round 1 hid behind os._exit: the engine's four bucket all-reduces were async.  Overlap
# with the backward pass comes from issuing the (synchronous) collective on a side stream instead - see
# MocoStepEngine._reduce_bucket.
def dist_all_reduce(tensor):
    import torch.distributed as dist
    dist.all_reduce(tensor)


def dist_all_reduce_pair(a, b):
    """all_reduce(a); all_reduce(b) as ONE collective where the backend can group them (RCCL: ncclGroupStart / End - one launch, one
    latency): the SyncBN statistics of encoder_q's and encoder_k's layer i, whose forward passes MoCo runs layer-locked (round 6).
    Synchronous, like every collective here (see above)."""
    import torch.distributed as dist
    if dist.get_backend() == "nccl" and a.is_cuda:
        with dist.distributed_c10d._coalescing_manager():
            dist.all_reduce(a)
            dist.all_reduce(b)
    else:
        dist.all_reduce(a)
        dist.all_reduce(b)


def bn_local_sums(x):
    """This rank's column sums (sum x, sum x^2: 2C doubles) of a channels-last activation - the statistics pass of a SyncBN whose
    all-reduce the caller issues itself (paired with another branch's: MoCo's layer-locked forward)."""
    _f32c(x, "x")
    c = x.shape[-1]
    m = x.numel() // c
    lib = L.lib()
    ws = _ws(lib.mi_colreduce_workspace_bytes(m, c), x.device, "colreduce")
    sums = torch.empty(2 * c, dtype=torch.float64, device=x.device)
    L.check(lib.mi_bn_stats(L.ptr(x), m, c, L.ptr(sums), L.ptr(ws), ws.numel(), L.stream()), "mi_bn_stats")
    return sums


def dist_all_gather(tensor_list, tensor):
    import torch.distributed as dist
    dist.all_gather(tensor_list, tensor)


def dist_all_gather_into(out, tensor):
    """all_gather along dim 0 into ONE contiguous tensor (synchronous, like the others: see above)."""
    import torch.distributed as dist
    dist.all_gather_into_tensor(out, tensor)


def _distributed():
    """True when the data-parallel exchanges (SyncBN sums, key all-gather, gradient all-reduce) have to be issued."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or FORCE_COLLECTIVES


BN_SMALL_MAX_ROWS = 4096      # == MI_BN_SMALL_MAX_ROWS


class _BNFn(torch.autograd.Function):
    """y = act(bn(x) + res); res (optional) is a residual branch added before the activation."""

    @staticmethod
    def forward(ctx, x, gamma, beta, mod, relu, res=None, pre=None, res_slot=None):
        ctx.res_slot = res_slot if res is not None else None
        shape = x.shape
        c = shape[-1]
        m = x.numel() // c
        lib = L.lib()
        dev = x.device
        distributed = mod.sync and _distributed()
        ctx.small = (mod.training or not mod.track_running_stats) and m <= BN_SMALL_MAX_ROWS and not distributed
        given_sums = None
        reduced = False
        if pre is not None and isinstance(pre[0], str):
            given_sums = pre[1]                           # this rank's column sums from the producing Linear's epilogue (SyncBN)
            reduced = pre[0] == "reduced"                 # ... or the GLOBAL sums: the caller has all-reduced them (paired with another branch's)
            pre = None
        if pre is not None:
            # (y, save) came out of the producing Linear's launch (linear_bn): nothing to run here
            if not ctx.small or res is not None:
                raise L.HipExtensionError("fused Linear+BatchNorm output handed to a BatchNorm that would not take the small path")
            y, save = pre
            ctx.count = float(m)
            ctx.train_stats = True
        else:
            y = torch.empty_like(x)
            save = torch.empty(2 * c, dtype=torch.float32, device=dev)
        if pre is not None:
            pass
        elif ctx.small:
            # one launch: statistics, running statistics, affine (+res, ReLU)
            track = mod.track_running_stats and mod.training
            L.check(lib.mi_bn_small_fwd(L.ptr(x), L.ptr(y), m, c, L.ptr(gamma), L.ptr(beta), mod.eps, mod.momentum,
                                        L.ptr(mod.running_mean if track else None),
                                        L.ptr(mod.running_var if track else None),
                                        L.ptr(mod.num_batches_tracked if track else None),
                                        L.ptr(save), L.ptr(res), int(relu), L.stream()), "mi_bn_small_fwd")
            ctx.count = float(m)
            ctx.train_stats = True
        elif mod.training or not mod.track_running_stats:
            if given_sums is not None:
                sums = given_sums
            else:
                ws = _ws(lib.mi_colreduce_workspace_bytes(m, c), dev, "colreduce")
                sums = torch.empty(2 * c, dtype=torch.float64, device=dev)
                L.check(lib.mi_bn_stats(L.ptr(x), m, c, L.ptr(sums), L.ptr(ws), ws.numel(), L.stream()), "mi_bn_stats")
            count = float(m)
            if mod.sync and _distributed():
                import torch.distributed as dist
                if not reduced:
                    dist_all_reduce(sums)                 # RCCL: 2*C doubles
                count = float(m) * _dist_world()
            track = mod.track_running_stats and mod.training
            L.check(lib.mi_bn_apply_fwd(L.ptr(x), L.ptr(y), m, c, L.ptr(sums), count, L.ptr(gamma), L.ptr(beta),
                                        mod.eps, mod.momentum,
                                        L.ptr(mod.running_mean if track else None),
                                        L.ptr(mod.running_var if track else None),
                                        L.ptr(mod.num_batches_tracked if track else None),
                                        L.ptr(save), L.ptr(res), int(relu), L.stream()), "mi_bn_apply_fwd")
            ctx.count = count
            ctx.train_stats = True
        else:
            L.check(lib.mi_bn_eval_fwd(L.ptr(x), L.ptr(y), m, c, L.ptr(mod.running_mean), L.ptr(mod.running_var),
                                       L.ptr(gamma), L.ptr(beta), mod.eps, L.ptr(save), L.ptr(res), int(relu),
                                       L.stream()), "mi_bn_eval_fwd")
            ctx.train_stats = False
        ctx.mod, ctx.relu, ctx.m, ctx.c = mod, relu, m, c
        ctx.has_res = res is not None
        ctx.save_for_backward(x, y if relu else None, save)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, save = ctx.saved_tensors
        mod, relu, m, c = ctx.mod, ctx.relu, ctx.m, ctx.c
        dy = dy.contiguous()
        lib = L.lib()
        dev = x.device
        if not ctx.train_stats:
            raise L.HipExtensionError("BatchNorm backward in eval mode is not on the hot path")
        dres = None
        want_dres = False
        if ctx.has_res:
            # the residual branch receives the gradient behind the activation; BN continues from it
            if relu and not ctx.small:
                want_dres = True                          # (written by the apply launch below: no masking launch in front of the two)
            else:
                if relu:
                    dy = relu_mask(dy, y)
                    relu = False
                dres = dy
        def hand_over(dres):
            # through the block's GradSlot when its first convolution adds the residual gradient in its own epilogue
            if ctx.res_slot is not None and dres is not None:
                ctx.res_slot.tensor = dres
                return None
            return dres
        if ctx.small:
            gamma = mod.weight
            dg = db = None
            acc_g = acc_b = False
            if gamma is not None and gamma.requires_grad:
                gt, acc_g = _grad_target(gamma)
                dg = torch.empty_like(gt) if acc_g else gt
                bt, acc_b = _grad_target(mod.bias)
                db = torch.empty_like(bt) if acc_b else bt
            dx = torch.empty_like(x)
            L.check(lib.mi_bn_small_bwd(L.ptr(dy), L.ptr(x), L.ptr(y), L.ptr(dx), m, c, L.ptr(save), L.ptr(gamma),
                                        int(relu), L.ptr(dg), L.ptr(db), L.stream()), "mi_bn_small_bwd")
            if acc_g:
                gamma.grad.add_(dg)
            if acc_b:
                mod.bias.grad.add_(db)
            return dx, None, None, None, None, hand_over(dres), None, None
        ws = _ws(lib.mi_colreduce_workspace_bytes(m, c), dev, "colreduce")
        sums = torch.empty(2 * c, dtype=torch.float64, device=dev)
        L.check(lib.mi_bn_bwd_reduce(L.ptr(dy), L.ptr(x), L.ptr(y), m, c, L.ptr(save), int(relu), L.ptr(sums),
                                     L.ptr(ws), ws.numel(), L.stream()), "mi_bn_bwd_reduce")
        gamma = mod.weight
        dg = db = None
        acc_g = acc_b = False
        if gamma is not None and gamma.requires_grad:
            gt, acc_g = _grad_target(gamma)
            dg = torch.empty_like(gt) if acc_g else gt
            bt, acc_b = _grad_target(mod.bias)
            db = torch.empty_like(bt) if acc_b else bt
        distributed = mod.sync and _distributed()
        if distributed:
            if dg is not None:
                # affine gradients come from the LOCAL sums (torch.nn.SyncBatchNorm does the same); the
                # data-parallel gradient averaging then treats them like every other parameter
                L.check(lib.mi_bn_param_grads(L.ptr(sums), c, L.ptr(dg), L.ptr(db), L.stream()), "mi_bn_param_grads")
            import torch.distributed as dist
            dist_all_reduce(sums)                      # dx needs the global sums
        dx = torch.empty_like(x)
        # single process: the same launch writes dgamma / dbeta from the sums
        if want_dres:
            dres = torch.empty_like(dy)
            L.check(lib.mi_bn_bwd_apply_res(L.ptr(dy), L.ptr(x), L.ptr(y), L.ptr(dx), L.ptr(dres), m, c, L.ptr(save), L.ptr(gamma),
                                            L.ptr(sums), ctx.count, L.ptr(None if distributed else dg),
                                            L.ptr(None if distributed else db), L.stream()), "mi_bn_bwd_apply_res")
        else:
            L.check(lib.mi_bn_bwd_apply(L.ptr(dy), L.ptr(x), L.ptr(y), L.ptr(dx), m, c, L.ptr(save), L.ptr(gamma),
                                        L.ptr(sums), ctx.count, int(relu), L.ptr(None if distributed else dg),
                                        L.ptr(None if distributed else db), L.stream()),
                    "mi_bn_bwd_apply")
        if acc_g:
            gamma.grad.add_(dg)
        if acc_b:
            mod.bias.grad.add_(db)
        return dx, None, None, None, None, hand_over(dres), None, None


class _BNReluPoolFn(torch.autograd.Function):
    """MaxPool3d(k, s, p)(relu(bn(x))) without materialising relu(bn(x)): the stem of the 3-D encoder."""

    @staticmethod
    def forward(ctx, x, gamma, beta, mod, k, stride, pad, given_sums=None, reduced=False):
        n, d, h, w, c = x.shape
        lib = L.lib()
        dev = x.device
        do, ho, wo = [(v + 2 * pad - k) // stride + 1 for v in (d, h, w)]
        y = torch.empty((n, do, ho, wo, c), dtype=torch.float32, device=dev)
        save = torch.empty(2 * c, dtype=torch.float32, device=dev)
        train = mod.training or not mod.track_running_stats
        arg = torch.empty((n, do, ho, wo, c), dtype=torch.uint8, device=dev) if (train and x.requires_grad) else None
        m = n * d * h * w
        count = float(m)
        sums = None
        if train:
            if given_sums is not None:
                sums = given_sums
            else:
                ws = _ws(lib.mi_colreduce_workspace_bytes(m, c), dev, "colreduce")
                sums = torch.empty(2 * c, dtype=torch.float64, device=dev)
                L.check(lib.mi_bn_stats(L.ptr(x), m, c, L.ptr(sums), L.ptr(ws), ws.numel(), L.stream()), "mi_bn_stats")
            if mod.sync and _distributed():
                import torch.distributed as dist
                if not reduced:                           # (reduced: the caller has all-reduced `given_sums`, paired with another branch's)
                    dist_all_reduce(sums)
                count = float(m) * _dist_world()
        track = mod.track_running_stats and mod.training
        use_running = not train
        L.check(lib.mi_bn_relu_maxpool3d_fwd(
            L.ptr(x), L.ptr(y), L.ptr(arg), n, d, h, w, c, k, stride, pad, L.ptr(sums), count, L.ptr(gamma), L.ptr(beta),
            mod.eps, mod.momentum, L.ptr(mod.running_mean if (track or use_running) else None),
            L.ptr(mod.running_var if (track or use_running) else None),
            L.ptr(mod.num_batches_tracked if track else None), L.ptr(save), L.stream()), "mi_bn_relu_maxpool3d_fwd")
        ctx.mod, ctx.geom, ctx.count, ctx.train = mod, (n, d, h, w, c, k, stride, pad), count, train
        ctx.save_for_backward(x, save, arg)
        return y

    @staticmethod
    def backward(ctx, dp):
        x, save, arg = ctx.saved_tensors
        if not ctx.train or arg is None:
            raise L.HipExtensionError("fused BatchNorm+ReLU+MaxPool backward needs a training-mode forward")
        mod = ctx.mod
        n, d, h, w, c, k, stride, pad = ctx.geom
        lib = L.lib()
        dev = x.device
        dp = dp.contiguous()
        m = n * d * h * w
        sums = torch.empty(2 * c, dtype=torch.float64, device=dev)
        gamma, beta = mod.weight, mod.bias
        # the pool's backward materialised once, then BatchNorm's two halves with the ReLU mask recomputed from x
        dy = torch.empty_like(x)
        L.check(lib.mi_maxpool3d_bwd(L.ptr(dp), L.ptr(arg), L.ptr(dy), n, d, h, w, c, k, stride, pad, L.stream()),
                "mi_maxpool3d_bwd")
        ws = _ws(lib.mi_colreduce_workspace_bytes(m, c), dev, "colreduce")
        L.check(lib.mi_bn_relu_bwd_reduce_x(L.ptr(dy), L.ptr(x), m, c, L.ptr(save), L.ptr(gamma), L.ptr(beta), L.ptr(sums),
                                            L.ptr(ws), ws.numel(), L.stream()), "mi_bn_relu_bwd_reduce_x")
        dg = db = None
        acc_g = acc_b = False
        if gamma is not None and gamma.requires_grad:
            gt, acc_g = _grad_target(gamma)
            dg = torch.empty_like(gt) if acc_g else gt
            bt, acc_b = _grad_target(beta)
            db = torch.empty_like(bt) if acc_b else bt
        distributed = mod.sync and _distributed()
        if distributed:
            if dg is not None:           # affine gradients from the LOCAL sums, like torch.nn.SyncBatchNorm
                L.check(lib.mi_bn_param_grads(L.ptr(sums), c, L.ptr(dg), L.ptr(db), L.stream()), "mi_bn_param_grads")
            import torch.distributed as dist
            dist_all_reduce(sums)
        dx = dy                                            # in place: each element is read, then written, by one thread
        L.check(lib.mi_bn_relu_bwd_apply_x(L.ptr(dy), L.ptr(x), L.ptr(dx), m, c, L.ptr(save), L.ptr(gamma), L.ptr(beta),
                                           L.ptr(sums), ctx.count, L.ptr(None if distributed else dg),
                                           L.ptr(None if distributed else db), L.stream()), "mi_bn_relu_bwd_apply_x")
        if acc_g:
            gamma.grad.add_(dg)
        if acc_b:
            beta.grad.add_(db)
        return dx, None, None, None, None, None, None, None, None


def bn_relu_maxpool3d(x, bn, k, stride, pad, sums=None, reduced=False):
    """maxpool3d(bn(x, relu=True), k, stride, pad) as one fused layer (bn: HipBatchNorm).  sums: the column sums of x and
    x^2 (2C doubles) where the producer of x already has them (the stem's epilogue) - the statistics pass is skipped."""
    return _BNReluPoolFn.apply(_f32c(x, "x"), bn.weight, bn.bias, bn, k, stride, pad, sums, reduced)


def linear_bn(x, lin, bn, relu=False):
    """bn(lin(x), relu) for a HipLinear followed by a HipBatchNorm over a batch of at most 64 rows in ONE launch
    (mi_linear_bn_fwd_f32: the product's tile holds every row of its columns, so the batch statistics are the workgroup's
    own); both autograd nodes stay, with their backward passes.  Anything else (more rows, eval mode, SyncBN across ranks)
    runs the two launches."""
    train = bn.training or not bn.track_running_stats
    if not (x.is_cuda and x.dim() == 2 and x.shape[0] <= 64 and train and PROFILE is None):
        return bn(lin(x), relu=relu)
    # SyncBN across ranks: only the statistics come out of the product's epilogue (this rank's column sums); the all-reduce and
    # the apply stay launches of their own
    lin._fuse_bn = (bn, relu, bool(bn.sync and _distributed()))
    try:
        xl = lin(x)
        pre = getattr(lin, "_bn_pre", None)
    finally:
        lin._fuse_bn = None
        lin._bn_pre = None
    return bn(xl, relu=relu, pre=pre)                 # (through the module: forward hooks keep firing)


def linear_with_local_sums(x, lin, bn):
    """(lin(x), this rank's column sums of it) for a SyncBN whose all-reduce the CALLER issues (MoCo's layer-locked forward pairs it
    with the other encoder's): the sums come out of the product's epilogue where linear_bn's would, else from a statistics pass."""
    train = bn.training or not bn.track_running_stats
    if x.is_cuda and x.dim() == 2 and x.shape[0] <= 64 and train and PROFILE is None:
        lin._fuse_bn = (bn, False, True)
        try:
            xl = lin(x)
            pre = getattr(lin, "_bn_pre", None)
        finally:
            lin._fuse_bn = None
            lin._bn_pre = None
        if pre is not None and pre[0] == "sums":
            return xl, pre[1]
        return xl, bn_local_sums(xl)
    xl = lin(x)
    return xl, bn_local_sums(xl)


class HipBatchNorm(nn.Module):
    """nn.BatchNorm3d / nn.BatchNorm1d over the last (channel) axis, optional fused ReLU."""

    def __init__(self, c, eps=1e-5, momentum=0.1, affine=True, track_running_stats=True):
        super().__init__()
        self.num_features, self.eps, self.momentum = c, eps, momentum
        self.affine, self.track_running_stats = affine, track_running_stats
        self.sync = False
        if affine:
            self.weight = nn.Parameter(torch.ones(c))
            self.bias = nn.Parameter(torch.zeros(c))
        else:
            self.register_parameter("weight", None)
            self.register_parameter("bias", None)
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))

    def forward(self, x, relu=False, res=None, pre=None, res_slot=None):
        return _BNFn.apply(_f32c(x, "x"), self.weight, self.bias, self, relu, res, pre, res_slot)


def convert_sync_batchnorm(module):
    """Counterpart of nn.SyncBatchNorm.convert_sync_batchnorm (moco_main.py:65-66): the same
    modules, with their per-channel sums all-reduced over RCCL."""
    for m in module.modules():
        if isinstance(m, HipBatchNorm):
            m.sync = True
    return module


# ------------------------------------------------------------------------------------------------
# pooling
# ------------------------------------------------------------------------------------------------
class _MaxPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, k, stride, pad):
        n, d, h, w, c = x.shape
        do, ho, wo = [(v + 2 * pad - k) // stride + 1 for v in (d, h, w)]
        y = torch.empty((n, do, ho, wo, c), dtype=torch.float32, device=x.device)
        arg = torch.empty((n, do, ho, wo, c), dtype=torch.uint8, device=x.device)
        L.check(L.lib().mi_maxpool3d_fwd(L.ptr(x), L.ptr(y), L.ptr(arg), n, d, h, w, c, k, stride, pad, L.stream()),
                "mi_maxpool3d_fwd")
        ctx.save_for_backward(arg)
        ctx.geom = (x.shape, k, stride, pad)
        return y

    @staticmethod
    def backward(ctx, dy):
        (arg,) = ctx.saved_tensors
        shape, k, stride, pad = ctx.geom
        n, d, h, w, c = shape
        dx = torch.empty(shape, dtype=torch.float32, device=dy.device)
        L.check(L.lib().mi_maxpool3d_bwd(L.ptr(dy.contiguous()), L.ptr(arg), L.ptr(dx), n, d, h, w, c, k, stride,
                                         pad, L.stream()), "mi_maxpool3d_bwd")
        return dx, None, None, None


def maxpool3d(x, k, stride, pad):
    return _MaxPoolFn.apply(_f32c(x, "x"), k, stride, pad)


class _AvgPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        n = x.shape[0]
        c = x.shape[-1]
        s = x.numel() // (n * c)
        y = torch.empty((n, c), dtype=torch.float32, device=x.device)
        L.check(L.lib().mi_avgpool_fwd(L.ptr(x), L.ptr(y), n, s, c, L.stream()), "mi_avgpool_fwd")
        ctx.shape = x.shape
        return y

    @staticmethod
    def backward(ctx, dy):
        shape = ctx.shape
        n, c = shape[0], shape[-1]
        s = 1
        for v in shape[1:-1]:
            s *= v
        dx = torch.empty(shape, dtype=torch.float32, device=dy.device)
        L.check(L.lib().mi_avgpool_bwd(L.ptr(dy.contiguous()), L.ptr(dx), n, s, c, L.stream()), "mi_avgpool_bwd")
        return dx


def global_avgpool(x):
    return _AvgPoolFn.apply(_f32c(x, "x"))


class _BNReluAvgPoolFn(torch.autograd.Function):
    """global_avgpool(relu(bn(x))) in the one launch of the small-M BatchNorm, forward and backward (the pooled sum runs in
    row order, the gradient is dy / V: the values of the separate avgpool kernels)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, mod, v):
        c = x.shape[-1]
        m = x.numel() // c
        lib = L.lib()
        y = torch.empty_like(x)
        pooled = torch.empty((x.shape[0], c), dtype=torch.float32, device=x.device)
        save = torch.empty(2 * c, dtype=torch.float32, device=x.device)
        track = mod.track_running_stats and mod.training
        L.check(lib.mi_bn_small_pool_fwd(L.ptr(x), L.ptr(y), L.ptr(pooled), m, c, v, L.ptr(gamma), L.ptr(beta), mod.eps,
                                         mod.momentum, L.ptr(mod.running_mean if track else None),
                                         L.ptr(mod.running_var if track else None),
                                         L.ptr(mod.num_batches_tracked if track else None), L.ptr(save), L.stream()),
                "mi_bn_small_pool_fwd")
        ctx.mod, ctx.m, ctx.c, ctx.v = mod, m, c, v
        ctx.save_for_backward(x, y, save)
        return pooled

    @staticmethod
    def backward(ctx, dp):
        x, y, save = ctx.saved_tensors
        mod, m, c, v = ctx.mod, ctx.m, ctx.c, ctx.v
        gamma = mod.weight
        dg = db = None
        acc_g = acc_b = False
        if gamma is not None and gamma.requires_grad:
            gt, acc_g = _grad_target(gamma)
            dg = torch.empty_like(gt) if acc_g else gt
            bt, acc_b = _grad_target(mod.bias)
            db = torch.empty_like(bt) if acc_b else bt
        dx = torch.empty_like(x)
        L.check(L.lib().mi_bn_small_pool_bwd(L.ptr(dp.contiguous()), L.ptr(x), L.ptr(y), L.ptr(dx), m, c, v, L.ptr(save),
                                             L.ptr(gamma), L.ptr(dg), L.ptr(db), L.stream()), "mi_bn_small_pool_bwd")
        if acc_g:
            gamma.grad.add_(dg)
        if acc_b:
            mod.bias.grad.add_(db)
        return dx, None, None, None, None


def bn_relu_global_avgpool(x, bn):
    """AdaptiveAvgPool3d(1)(relu(bn(x))) -> (N, C): one launch each way when the one-launch BatchNorm applies (training
    statistics, <= BN_SMALL_MAX_ROWS rows, no SyncBN exchange, a power-of-two number of voxels per sample <= 64);
    otherwise the separate kernels."""
    _f32c(x, "x")
    n, c = x.shape[0], x.shape[-1]
    m = x.numel() // c
    v = m // max(n, 1)
    fused = (x.is_cuda and (bn.training or not bn.track_running_stats) and m <= BN_SMALL_MAX_ROWS and c % 4 == 0
             and not (bn.sync and _distributed()) and 1 <= v <= 64 and (v & (v - 1)) == 0 and v * n == m)
    if not fused:
        return global_avgpool(bn(x, relu=True))
    return _BNReluAvgPoolFn.apply(x, bn.weight, bn.bias, bn, v)


# ------------------------------------------------------------------------------------------------
# BasicBlock (moco_encoder_3d.py:55-84): conv-ReLU-conv (+residual) -ReLU, no BN, with the ReLU
# derivatives fused into the data-gradient epilogues
# ------------------------------------------------------------------------------------------------
RELU_TAP = None      # tests set a dict: post-ReLU activations of a forward pass (block: id(module) -> (mid, out); the encoder
                     # adds "feature_3d", "proj.1", "proj.4"), copies - so that a float64 oracle can take the SAME branch of the
                     # piecewise-linear network the GPU took where a unit sits within rounding of zero


class _BasicBlockFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w1, w2, wds, blk, mask_dx=False, dout_masked=False):
        s = blk.stride
        front = conv_fwd_s2_block(x, w1, wds) if (s == 2 and wds is not None) else None
        if front is not None:
            hmid, r = front                                  # stride-2 front: conv1 + ReLU and the shortcut in one launch
        else:
            hmid = conv_fwd(x, w1, 3, s, 1, None, True)
            r = conv_fwd(x, wds, 1, s, 0) if wds is not None else x
        out = conv_fwd(hmid, w2, 3, 1, 1, r, True)
        if RELU_TAP is not None:                             # (tests: the ReLU decisions of this forward pass)
            RELU_TAP[id(blk)] = (hmid.detach().clone(), out.detach().clone())
        ctx.blk = blk
        ctx.mask_dx, ctx.dout_masked = mask_dx, dout_masked
        ctx.save_for_backward(x, hmid, out)
        ctx.x_needs_grad = x.requires_grad
        return out

    @staticmethod
    def backward(ctx, dout):
        x, hmid, out = ctx.saved_tensors
        blk = ctx.blk
        s = blk.stride
        # through the block's last ReLU - unless the consumer of `out` already did it in its data-gradient epilogue
        d2 = dout.contiguous() if ctx.dout_masked else relu_mask(dout.contiguous(), out)
        xmask = x if ctx.mask_dx else None                           # ... as this block does for ITS producer
        if blk.conv2.weight.requires_grad:
            conv_wgrad_into(hmid, d2, blk.conv2.weight, 3, 1, 1)
        dh = conv_dgrad(d2, blk.conv2.weight, hmid.shape, 3, 1, 1, None, hmid)   # * (hmid > 0) fused
        if blk.conv1.weight.requires_grad:
            conv_wgrad_into(x, dh, blk.conv1.weight, 3, s, 1)
        dx = None
        ds = blk.downsample
        if ds is not None:
            if ds[0].weight.requires_grad:
                conv_wgrad_into(x, d2, ds[0].weight, 1, s, 0)
            if ctx.x_needs_grad:
                # stride-2 block front: both data gradients (+ the ReLU mask) in one launch where the shape is the encoder's
                dx = conv_dgrad_s2_block(dh, d2, blk.conv1.weight, ds[0].weight, x.shape, None, xmask) if s == 2 else None
                if dx is None:
                    dres = conv_dgrad(d2, ds[0].weight, x.shape, 1, s, 0)
                    dx = conv_dgrad(dh, blk.conv1.weight, x.shape, 3, s, 1, dres, xmask)
        elif ctx.x_needs_grad:
            dx = conv_dgrad(dh, blk.conv1.weight, x.shape, 3, s, 1, d2, xmask)   # + identity branch fused
        return dx, None, None, None, None, None, None


def basic_block(x, blk, mask_dx=False, dout_masked=False):
    """mask_dx: x is the output of a ReLU whose derivative its producer leaves to this block - the returned gradient is
    (dx) * (x > 0), applied in the data-gradient epilogue (x > 0 exactly where the producer's pre-activation was).
    dout_masked: the ONE consumer of this block's output does the same for this block's last ReLU, so the backward skips
    its own mask launch.  The caller (the encoder's trunk) pairs the two flags; defaults keep the block self-contained."""
    wds = blk.downsample[0].weight if blk.downsample is not None else None
    return _BasicBlockFn.apply(_f32c(x, "x"), blk.conv1.weight, blk.conv2.weight, wds, blk, mask_dx, dout_masked)


# ------------------------------------------------------------------------------------------------
# contrastive head: normalise, logits, cross-entropy
# ------------------------------------------------------------------------------------------------
class _L2NormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        b, c = x.shape
        y = torch.empty_like(x)
        inv = torch.empty(b, dtype=torch.float32, device=x.device)
        L.check(L.lib().mi_l2norm_fwd(L.ptr(x), L.ptr(y), L.ptr(inv), b, c, L.stream()), "mi_l2norm_fwd")
        ctx.save_for_backward(y, inv)
        return y

    @staticmethod
    def backward(ctx, dy):
        y, inv = ctx.saved_tensors
        b, c = y.shape
        dx = torch.empty_like(y)
        L.check(L.lib().mi_l2norm_bwd(L.ptr(dy.contiguous()), L.ptr(y), L.ptr(inv), L.ptr(dx), b, c, L.stream()),
                "mi_l2norm_bwd")
        return dx


def l2_normalize(x):
    return _L2NormFn.apply(_f32c(x, "x"))


class _MocoLogitsFn(torch.autograd.Function):
    """logits = cat([q.k, q @ queue], 1) / T   (models/moco.py:130-138); gradient w.r.t. q only."""

    @staticmethod
    def forward(ctx, q, k, queue, T):
        b, c = q.shape
        r = queue.shape[1]
        logits = torch.empty((b, r + 1), dtype=torch.float32, device=q.device)
        L.check(L.lib().mi_moco_logits_fwd(L.ptr(q), L.ptr(k), L.ptr(queue), L.ptr(logits), b, c, r, float(T),
                                           L.stream()), "mi_moco_logits_fwd")
        ctx.save_for_backward(k, queue.clone())      # the queue is overwritten by the enqueue below
        ctx.T = float(T)
        return logits

    @staticmethod
    def backward(ctx, dl):
        k, queue = ctx.saved_tensors
        b, c = k.shape
        r = queue.shape[1]
        dq = torch.empty_like(k)
        L.check(L.lib().mi_moco_logits_bwd(L.ptr(dl.contiguous()), L.ptr(k), L.ptr(queue), L.ptr(dq), b, c, r,
                                           ctx.T, L.stream()), "mi_moco_logits_bwd")
        return dq, None, None, None


def moco_logits(q, k, queue, T):
    return _MocoLogitsFn.apply(_f32c(q, "q"), _f32c(k, "k"), _f32c(queue, "queue"), T)


class _MocoLogitsNormFn(torch.autograd.Function):
    """(logits, k_hat) from the un-normalised projections: normalize(q), normalize(k) and the logits in one launch; the
    backward is logits_bwd + l2norm_bwd on the saved q_hat / 1/|q| (gradient w.r.t. q_raw only).
    queue_stable: the caller enqueues only after the backward pass, so the queue needs no copy."""

    @staticmethod
    def forward(ctx, q_raw, k_raw, queue, T, queue_stable):
        b, c = q_raw.shape
        r = queue.shape[1]
        dev = q_raw.device
        logits = torch.empty((b, r + 1), dtype=torch.float32, device=dev)
        q_hat = torch.empty_like(q_raw)
        k_hat = torch.empty_like(q_raw)
        q_inv = torch.empty(b, dtype=torch.float32, device=dev)
        L.check(L.lib().mi_moco_logits_norm_fwd(L.ptr(q_raw), L.ptr(k_raw), L.ptr(queue), L.ptr(logits), L.ptr(q_hat),
                                                L.ptr(q_inv), L.ptr(k_hat), b, c, r, float(T), L.stream()),
                "mi_moco_logits_norm_fwd")
        ctx.save_for_backward(k_hat, queue if queue_stable else queue.clone(), q_hat, q_inv)
        ctx.T = float(T)
        ctx.mark_non_differentiable(k_hat)
        ctx.set_materialize_grads(False)        # (no zeros(B, C) launch for the key output's absent gradient)
        return logits, k_hat

    @staticmethod
    def backward(ctx, dl, _dk):
        if dl is None:
            return None, None, None, None, None
        k_hat, queue, q_hat, q_inv = ctx.saved_tensors
        b, c = k_hat.shape
        r = queue.shape[1]
        lib = L.lib()
        dq = torch.empty_like(k_hat)
        L.check(lib.mi_moco_logits_bwd(L.ptr(dl.contiguous()), L.ptr(k_hat), L.ptr(queue), L.ptr(dq), b, c, r, ctx.T,
                                       L.stream()), "mi_moco_logits_bwd")
        dx = torch.empty_like(dq)
        L.check(lib.mi_l2norm_bwd(L.ptr(dq), L.ptr(q_hat), L.ptr(q_inv), L.ptr(dx), b, c, L.stream()), "mi_l2norm_bwd")
        return dx, None, None, None, None


def moco_logits_normalized(q_raw, k_raw, queue, T, queue_stable=False):
    """models/moco.py:113-138 from the encoders' raw outputs; returns (logits, normalize(k))."""
    return _MocoLogitsNormFn.apply(_f32c(q_raw, "q"), _f32c(k_raw, "k"), _f32c(queue, "queue"), T, bool(queue_stable))


_CE0_COUNTERS = {}       # device -> (int32 tensor of 64 self-resetting arrival counters, {stream id: slot})


def _ce0_counter(device):
    """The arrival counter of this stream's cross-entropy launches: one 4-byte word per stream out of a per-device table that
    is zeroed ONCE, when it is created (a per-stream workspace would be created - and its clearing pass captured - inside a
    hipGraph capture, whose capture stream is a new one: a fill launch in every replay).  Launches on different streams never
    share a word; the kernel leaves its word zero."""
    key = (device.type, device.index)
    ent = _CE0_COUNTERS.get(key)
    if ent is None:
        ent = (torch.zeros(64 * 64, dtype=torch.int32, device=device), {})      # 64 words, 256 bytes apart
        _CE0_COUNTERS[key] = ent
    table, slots = ent
    sid = torch.cuda.current_stream(device).cuda_stream
    slot = slots.get(sid)
    if slot is None:
        if len(slots) >= 64:
            # never alias: two live streams on one arrival word mix their tickets (a wrong or missing loss write)
            raise L.HipExtensionError("cross_entropy_label0 has been launched on 64 distinct HIP streams of %s; "
                                    "hipops.reset_ce0_counters() releases the words of streams that no longer exist" % (device,))
        slot = len(slots)
        slots[sid] = slot
    return table[slot * 64:]


def reset_ce0_counters(device=None):
    """Forget the stream -> arrival-word table (all devices, or one).  Call it only with no cross-entropy launch in flight
    and outside a hipGraph capture; graphs captured before keep the addresses of the old table alive through their tensors."""
    for key in list(_CE0_COUNTERS):
        if device is None or key == (device.type, device.index):
            del _CE0_COUNTERS[key]


class _CELabel0Fn(torch.autograd.Function):
    """forward = one launch (a workgroup per row, the last one takes the mean) that also drops the loss into `out` (the
    engine's loss buffer: no copy afterwards); backward = one launch that reads the upstream gradient on the device."""

    @staticmethod
    def forward(ctx, logits, out):
        b, n = logits.shape
        loss = torch.empty((), dtype=torch.float32, device=logits.device)
        rows = torch.empty(2 * b, dtype=torch.float32, device=logits.device)          # row losses | row log-sum-exps
        word = _ce0_counter(logits.device)
        rc = L.lib().mi_ce_label0_fwd(L.ptr(logits), L.ptr(loss), L.ptr(out), L.ptr(rows), L.ptr(rows[b:]), b, n,
                                      L.ptr(word), L.stream())
        if rc != 0 and not torch.cuda.is_current_stream_capturing():
            word[:1].zero_()                           # a launch that failed midway must not leave its tickets behind
        L.check(rc, "mi_ce_label0_fwd")
        ctx.save_for_backward(logits, rows)
        return loss

    @staticmethod
    def backward(ctx, g):
        logits, rows = ctx.saved_tensors
        b, n = logits.shape
        dl = torch.empty_like(logits)
        L.check(L.lib().mi_ce_label0_bwd(L.ptr(logits), L.ptr(rows[b:]), L.ptr(g.contiguous()), L.ptr(dl), b, n, L.stream()),
                "mi_ce_label0_bwd")
        return dl, None


def cross_entropy_label0(logits, out=None):
    """nn.CrossEntropyLoss()(logits, zeros) (trains/tomo_moco_trainer.py:52,73).  `out`: a 0-d fp32 device tensor that
    receives the loss (and is returned)."""
    return _CELabel0Fn.apply(_f32c(logits, "logits"), out)


class _RowDotMeanFn(torch.autograd.Function):
    """mean_b(a_b . b_b); gradient w.r.t. a only (b is the detached target)."""

    @staticmethod
    def forward(ctx, a, b):
        bsz, c = a.shape
        out = torch.empty((), dtype=torch.float32, device=a.device)
        L.check(L.lib().mi_rowdot_mean_fwd(L.ptr(a), L.ptr(b), L.ptr(out), bsz, c, L.stream()), "mi_rowdot_mean_fwd")
        ctx.save_for_backward(b)
        return out

    @staticmethod
    def backward(ctx, g):
        (b,) = ctx.saved_tensors
        da = torch.empty_like(b)
        L.check(L.lib().mi_rowdot_mean_bwd(L.ptr(b), L.ptr(g.contiguous()), L.ptr(da), b.shape[0], b.shape[1],
                                           L.stream()), "mi_rowdot_mean_bwd")
        return da, None


def rowdot_mean(a, b):
    return _RowDotMeanFn.apply(_f32c(a, "a"), _f32c(b, "b"))


def column_std_mean(x):
    """torch.std(x, 0).mean() (the SimSiam `output_std` monitor)."""
    out = torch.empty((), dtype=torch.float32, device=x.device)
    L.check(L.lib().mi_column_std_mean(L.ptr(_f32c(x, "x")), L.ptr(out), x.shape[0], x.shape[1], L.stream()),
            "mi_column_std_mean")
    return out


# ------------------------------------------------------------------------------------------------
# optimiser-side passes
# ------------------------------------------------------------------------------------------------
def ema_update_(k_flat, q_flat, m):
    _bump_weight_epoch()                 # a raw-pointer write: torch's version counters do not see it
    L.check(L.lib().mi_ema_update(L.ptr(k_flat), L.ptr(q_flat), float(m), k_flat.numel(), L.stream()),
            "mi_ema_update")


def sgd_step_(p_flat, g_flat, lr, weight_decay=0.0, lr_dev=None, grad_scale=1.0):
    """p -= lr * (grad_scale * g + wd * p); grad_scale = 1 / world when g is the SUM of the ranks' gradients."""
    _bump_weight_epoch()
    L.check(L.lib().mi_sgd_step(L.ptr(p_flat), L.ptr(g_flat), L.ptr(lr_dev), float(lr), float(weight_decay),
                                float(grad_scale), p_flat.numel(), L.stream()), "mi_sgd_step")


def sgd_step2_(p_flat, g_flat, g2_flat, lr, weight_decay=0.0, lr_dev=None, grad_scale=1.0):
    """p -= lr * (grad_scale * (g + g2) + wd * p): two gradient arenas (one per view of a two-view model)."""
    _bump_weight_epoch()
    L.check(L.lib().mi_sgd_step2(L.ptr(p_flat), L.ptr(g_flat), L.ptr(g2_flat), L.ptr(lr_dev), float(lr), float(weight_decay),
                                 float(grad_scale), p_flat.numel(), L.stream()), "mi_sgd_step2")


def scalar_accumulate_(sums, *scalars):
    """sums[i] += scalars[i] (0-d / 1-element f32 device tensors, at most four) in one launch."""
    if len(scalars) > 4 or sums.numel() < len(scalars):
        raise L.HipExtensionError("scalar_accumulate_: at most four scalars, sums at least as long")
    ptrs = [L.ptr(_f32c(t, "scalar")) for t in scalars] + [None] * (4 - len(scalars))
    L.check(L.lib().mi_scalar_accumulate(L.ptr(sums), *ptrs, L.stream()), "mi_scalar_accumulate")


def copy_pair_(dst0, src0, dst1, src1):
    """dst0.copy_(src0); dst1.copy_(src1) - as ONE launch when the four are contiguous f32 device tensors of one size."""
    ts = (dst0, src0, dst1, src1)
    if (all(t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() for t in ts) and src0.shape == dst0.shape
            and src1.shape == dst1.shape and dst0.numel() == dst1.numel() and dst0.numel() % 4 == 0
            and all(t.data_ptr() % 16 == 0 for t in ts)):
        L.check(L.lib().mi_copy_pair_f32(L.ptr(dst0), L.ptr(src0), L.ptr(dst1), L.ptr(src1), dst0.numel(), L.stream()),
                "mi_copy_pair_f32")
        return
    dst0.copy_(src0)
    dst1.copy_(src1)


def queue_enqueue_(queue, queue_ptr, keys):
    c, r = queue.shape
    b = keys.shape[0]
    if r % b != 0:
        raise AssertionError("queue size must be a multiple of the batch (models/moco.py:47)")
    L.check(L.lib().mi_queue_enqueue(L.ptr(queue), L.ptr(queue_ptr), L.ptr(keys), b, c, r, L.stream()),
            "mi_queue_enqueue")


# ------------------------------------------------------------------------------------------------
# detector network glue (models/networks/unet.py): ceil-mode max-pool, 2x2 transposed conv, concat, bias, z head
# ------------------------------------------------------------------------------------------------
class _MaxPool2dCeilFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, k):
        n, h, w, c = x.shape
        ho, wo = (h + k - 1) // k, (w + k - 1) // k
        y = torch.empty((n, ho, wo, c), dtype=torch.float32, device=x.device)
        arg = torch.empty((n, ho, wo, c), dtype=torch.uint8, device=x.device) if x.requires_grad else None
        L.check(L.lib().mi_maxpool2d_ceil_fwd(L.ptr(x), L.ptr(y), L.ptr(arg), n, h, w, c, k, L.stream()),
                "mi_maxpool2d_ceil_fwd")
        ctx.save_for_backward(arg)
        ctx.geom = (x.shape, k)
        return y

    @staticmethod
    def backward(ctx, dy):
        (arg,) = ctx.saved_tensors
        shape, k = ctx.geom
        n, h, w, c = shape
        dx = torch.empty(shape, dtype=torch.float32, device=dy.device)
        L.check(L.lib().mi_maxpool2d_ceil_bwd(L.ptr(dy.contiguous()), L.ptr(arg), L.ptr(dx), n, h, w, c, k, L.stream()),
                "mi_maxpool2d_ceil_bwd")
        return dx, None


def maxpool2d_ceil(x, k=2):
    """nn.MaxPool2d(k, ceil_mode=True) on (N,H,W,C)."""
    return _MaxPool2dCeilFn.apply(_f32c(x, "x"), k)


def _colsum_into(dy2d, param):
    """param.grad (+)= column sums of dy2d (M, C)."""
    m, c = dy2d.shape
    lib = L.lib()
    g, acc = _grad_target(param)
    tgt = torch.empty_like(g) if acc else g
    ws = _ws(lib.mi_colreduce_workspace_bytes(m, c), dy2d.device, "colreduce")
    sums = torch.empty(2 * c, dtype=torch.float64, device=dy2d.device)
    L.check(lib.mi_colsum(L.ptr(dy2d), m, c, L.ptr(tgt), L.ptr(sums), L.ptr(ws), ws.numel(), L.stream()), "mi_colsum")
    if acc:
        g.add_(tgt)


class _ConvT2x2Fn(torch.autograd.Function):
    """nn.ConvTranspose2d(ci, co, 2, stride=2) cropped to (ho, wo): 1x1 implicit GEMM to 4*co columns + pixel shuffle."""

    @staticmethod
    def forward(ctx, x, w, bias, mod, ho, wo, inference=False):
        n, h, wd, ci = x.shape
        co = mod.co
        t = conv_fwd(x, mod.gemm_view(), 1, 1, 0, owner=mod.weight, inference=inference)
        y = torch.empty((n, ho, wo, co), dtype=torch.float32, device=x.device)
        L.check(L.lib().mi_shuffle2x2_fwd(L.ptr(t), L.ptr(bias), L.ptr(y), n, h, wd, co, ho, wo, L.stream()),
                "mi_shuffle2x2_fwd")
        ctx.save_for_backward(x)
        ctx.mod, ctx.out_hw = mod, (ho, wo)
        ctx.x_needs_grad = x.requires_grad
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        mod = ctx.mod
        n, h, wd, ci = x.shape
        co = mod.co
        ho, wo = ctx.out_hw
        dy = dy.contiguous()
        dt = torch.empty((n, h, wd, 4 * co), dtype=torch.float32, device=dy.device)
        L.check(L.lib().mi_shuffle2x2_bwd(L.ptr(dy), L.ptr(dt), n, h, wd, co, ho, wo, L.stream()), "mi_shuffle2x2_bwd")
        if mod.bias is not None and mod.bias.requires_grad:
            _colsum_into(dy.view(-1, co), mod.bias)
        if mod.weight.requires_grad:
            # the gradient tensor has the parameter's strides, i.e. the GEMM layout [ci][4*co]
            g, acc = _grad_target(mod.weight)
            tgt = torch.empty_like(g) if acc else g
            lib = L.lib()
            ws = _ws(lib.mi_convnd_workspace_bytes(n, 1, h, wd, ci, 4 * co, 1, 1, 1, 1, 0, 0, 0), x.device, "conv")
            L.check(lib.mi_convnd_wgrad_f32(L.ptr(x), L.ptr(dt), L.ptr(tgt), n, 1, h, wd, ci, 4 * co, 1, 1, 1, 1, 0, 0, 0,
                                            L.ptr(ws), ws.numel(), L.stream()), "mi_convnd_wgrad_f32")
            if acc:
                g.add_(tgt)
        dx = conv_dgrad(dt, mod.gemm_view(), x.shape, 1, 1, 0) if ctx.x_needs_grad else None
        return dx, None, None, None, None, None, None


class HipConvTranspose2x2(nn.Module):
    """nn.ConvTranspose2d(ci, co, kernel_size=2, stride=2): weight logical (ci, co, 2, 2) over storage [ci][a][b][co]."""

    def __init__(self, ci, co):
        super().__init__()
        self.ci, self.co = ci, co
        self.weight = nn.Parameter(torch.empty(ci, 2, 2, co).permute(0, 3, 1, 2))
        self.bias = nn.Parameter(torch.zeros(co))
        with torch.no_grad():
            bound = 1.0 / (co * 4) ** 0.5             # torch's fan_in of a transposed conv weight = size(1) * k*k
            self.weight.uniform_(-bound, bound)
            self.bias.uniform_(-bound, bound)

    def gemm_view(self):
        """the same storage as a (4*co, ci, 1, 1) convolution weight in kernel layout"""
        return torch.as_strided(self.weight, (4 * self.co, self.ci, 1, 1), (1, 4 * self.co, 4 * self.co * self.ci,
                                                                            4 * self.co * self.ci))

    def forward(self, x, ho=None, wo=None):
        n, h, w, _ = x.shape
        return _ConvT2x2Fn.apply(_f32c(x, "x"), self.weight, self.bias, self, 2 * h if ho is None else ho,
                                 2 * w if wo is None else wo, inference_mode())


class _ConcatFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        ca, cb = a.shape[-1], b.shape[-1]
        m = a.numel() // ca
        out = torch.empty(a.shape[:-1] + (ca + cb,), dtype=torch.float32, device=a.device)
        L.check(L.lib().mi_concat_channels(L.ptr(a), ca, L.ptr(b), cb, L.ptr(out), m, L.stream()), "mi_concat_channels")
        ctx.shapes = (a.shape, b.shape)
        return out

    @staticmethod
    def backward(ctx, dout):
        sa, sb = ctx.shapes
        da = torch.empty(sa, dtype=torch.float32, device=dout.device)
        db = torch.empty(sb, dtype=torch.float32, device=dout.device)
        L.check(L.lib().mi_split_channels(L.ptr(dout.contiguous()), L.ptr(da), sa[-1], L.ptr(db), sb[-1],
                                          da.numel() // sa[-1], L.stream()), "mi_split_channels")
        return da, db


def concat_channels(a, b):
    """torch.cat((a, b), 1) of the reference (channel axis) on channels-last tensors."""
    return _ConcatFn.apply(_f32c(a, "a"), _f32c(b, "b"))


class _BiasAddFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, bias, holder):
        y = x.clone()
        c = x.shape[-1]
        L.check(L.lib().mi_bias_add(L.ptr(y), L.ptr(bias), y.numel() // c, c, L.stream()), "mi_bias_add")
        ctx.holder = holder
        return y

    @staticmethod
    def backward(ctx, dy):
        b = ctx.holder.bias
        if b.requires_grad:
            _colsum_into(dy.contiguous().view(-1, dy.shape[-1]), b)
        return dy, None, None


def bias_add(x, holder):
    """x + holder.bias over the channel axis; the bias gradient goes to holder.bias.grad."""
    return _BiasAddFn.apply(_f32c(x, "x"), holder.bias, holder)


class _ZHeadFn(torch.autograd.Function):
    """nn.Conv3d(C, K, (3,1,1), padding=(1,0,0), bias=False) with K <= 4 on (N,D,H,W,C)."""

    @staticmethod
    def forward(ctx, x, w, mod):
        n, d, h, wd, c = x.shape
        k = mod.k_out
        y = torch.empty((n, d, h, wd, k), dtype=torch.float32, device=x.device)
        L.check(L.lib().mi_zhead_fwd(L.ptr(x), L.ptr(w), L.ptr(y), n, d, h * wd, c, k, L.stream()), "mi_zhead_fwd")
        ctx.save_for_backward(x)
        ctx.mod = mod
        ctx.x_needs_grad = x.requires_grad
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        mod = ctx.mod
        n, d, h, wd, c = x.shape
        k = mod.k_out
        lib = L.lib()
        dy = dy.contiguous()
        dx = torch.empty_like(x) if ctx.x_needs_grad else None
        tgt, g, acc = None, None, False
        if mod.weight.requires_grad:
            g, acc = _grad_target(mod.weight)
            tgt = torch.empty_like(g) if acc else g
        ws = _ws(lib.mi_zhead_bwd_workspace_bytes(n, d, h * wd, c, k), x.device, "zhead")
        L.check(lib.mi_zhead_bwd(L.ptr(x), L.ptr(mod.weight), L.ptr(dy), L.ptr(dx), L.ptr(tgt), n, d, h * wd, c, k,
                                 L.ptr(ws), ws.numel(), L.stream()), "mi_zhead_bwd")
        if acc:
            g.add_(tgt)
        return dx, None, None


class HipZHead(nn.Module):
    """The (3,1,1) head with <= 4 outputs: weight logical (K, C, 3, 1, 1) over storage [3][C][K]."""

    def __init__(self, c, k_out):
        super().__init__()
        self.c, self.k_out = c, k_out
        self.weight = nn.Parameter(torch.empty(3, 1, 1, c, k_out).permute(4, 3, 0, 1, 2))
        with torch.no_grad():
            self.weight.normal_(std=0.001)

    def forward(self, x):
        return _ZHeadFn.apply(_f32c(x, "x"), self.weight, self)

