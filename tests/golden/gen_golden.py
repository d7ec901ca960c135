"""Generate golden vectors by IMPORTING the reference (read-only at /root/reference) in the build
container.  Run:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_golden.py
The reference never travels; only the small .npz / .json outputs written next to this script do.
Inputs are regenerated from seeds by the tests (cet_pick_amd.synthetic), so fixtures hold outputs
(plus small inputs where they are tiny).
"""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, "/root/reference")
sys.dont_write_bytecode = True


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


# inert stubs for packages the hot functions never touch
_stub("cv2")
tv = _stub("torchvision")
tvt = _stub("torchvision.transforms")
tvf = _stub("torchvision.transforms.functional", InterpolationMode=type("InterpolationMode", (), {}))
tv.transforms = tvt
tvt.functional = tvf
_stub("skimage")
_stub("skimage.transform", rescale=None)


class _MrcHandle:
    """stand-in for mrcfile.open(path, permissive=True): context manager exposing .data (the array)"""
    def __init__(self, path, **kw):
        from cet_pick_amd.utils.mrc import open_data
        self.data = open_data(path)

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


_stub("mrcfile", open=_MrcHandle)

from cet_pick.models import decode as R_decode            # noqa: E402
from cet_pick.models.utils import _sigmoid as R_sigmoid   # noqa: E402
from cet_pick.utils import image as R_image               # noqa: E402
from cet_pick.models.networks import moco_encoder_3d as R_enc3d  # noqa: E402
from cet_pick.models import moco as R_moco                # noqa: E402
from cet_pick.utils import utils as R_utils               # noqa: E402

from cet_pick_amd.synthetic import make_tomo, make_logits, seeded_state_dict, losses_inputs  # noqa: E402


def gen_loader():
    """utils/loader.py load_rec / preprocess / quantize and utils/mrc.py write -> loader_small.npz, ref_written.mrc"""
    import tempfile
    from cet_pick.utils import loader as R_loader
    from cet_pick.utils import mrc as R_mrc
    from cet_pick_amd.utils import mrc as my_mrc
    rng = np.random.default_rng(23)
    vol = (rng.standard_normal((10, 12, 9)) * 3.0 + 40.0).astype(np.float32)
    vol[2:5, 3:8, 2:6] += 9.0
    out = {"vol": vol}
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "v.mrc")
        my_mrc.write(path, vol)
        for order in ("xyz", "xzy", "yxz", "zxy"):
            for comp in (False, True):
                out[f"load_{order}_{int(comp)}"] = R_loader.load_rec(path, order=order, compress=comp)
            out[f"load_tilt_{order}"] = R_loader.load_rec(path, order=order, compress=True, is_tilt=True)
        v16 = np.round(vol * 10).astype(np.int16)
        my_mrc.write(path, v16)
        out["vol_i16"] = v16
        out["load_i16_xzy_1"] = R_loader.load_rec(path, order="xzy", compress=True)
        # the reference's own writer / reader: a file fixture + what its parser returns
        ref_path = os.path.join(HERE, "ref_written.mrc")
        R_mrc.write(ref_path, vol[:4, :5, :6].copy())
        arr, hdr = R_mrc.parse_mrc(ref_path)
        out["ref_written_data"] = arr
        out["ref_written_fields"] = np.array([hdr.fields[k] for k in ("nx", "ny", "nz", "mode", "mapc", "mapr", "maps", "ispg", "next")])
        my_mrc.write(path, vol)
        arr2, hdr2 = R_mrc.parse_mrc(path)      # the reference reads what this build writes
        assert np.array_equal(arr2, vol) and hdr2.fields["nx"] == 9
    z = R_loader.load_rec.__globals__["np"].asarray(out["load_xzy_0"])
    out["pre_0"] = R_loader.preprocess(z, denoise=0)
    out["pre_dn"] = R_loader.preprocess(z, denoise=1.0)
    out["quant"] = R_loader.quantize(z)
    out["quant_33"] = R_loader.quantize(z, mi=-3, ma=3)
    out["quant_auto"] = R_loader.quantize(z, mi=None, ma=None)
    save("loader_small.npz", **out)


def gen_unet():
    """a22: TomoConvUNet (unet_small.py:30-97) eval-mode forward, unet_4, heads {'hm': 1, 'proj': 32}."""
    from cet_pick.models.networks import unet_small as RU
    heads = {"hm": 1, "proj": 32}
    net = RU.TomoConvUNet(4, heads, 32, 3)
    net.load_state_dict(seeded_state_dict(net, seed=321))
    net.eval()
    g = torch.Generator().manual_seed(7)
    res = {}
    for tag, shape in (("a", (1, 8, 64, 64)), ("odd", (1, 5, 52, 44)), ("b2", (2, 4, 48, 48))):
        x = torch.randn(shape, generator=g)
        with torch.no_grad():
            out = net(x)[0]
        res[f"x_{tag}"] = x.numpy()
        res[f"hm_{tag}"] = out["hm"].numpy()
        pr = out["proj"].numpy()
        res[f"proj_{tag}"] = pr[:, :, :, ::3, ::3].copy()       # subsampled: keeps the fixture small
    save("unet4.npz", **res)
    path = os.path.join(HERE, "ckpt_keys.json")
    keys = json.load(open(path))
    keys["unet_4"] = {k: list(v.shape) for k, v in net.state_dict().items()}
    json.dump(keys, open(path, "w"), indent=0)


def gen_simsiam():
    """a3: TomoResClassifier (simsiam_model.py:159-440), arch 'simsiam': two-view train forward/backward + eval."""
    from cet_pick.models.networks import simsiam_model as RS
    heads = {"proj": 256, "pred": 256}
    net = RS.TomoResClassifier(RS.BasicBlock, [2, 2, 2, 2], heads, 0)
    net.load_state_dict(seeded_state_dict(net, seed=319))
    g = torch.Generator().manual_seed(6)
    x1 = torch.randn(4, 5, 40, 40, generator=g)
    x2 = x1.flip(3) + 0.1 * torch.randn(4, 5, 40, 40, generator=g)
    net.train()
    out = net(x1, x2)
    p1, z1, p2, z2 = out[0]["pred"], out[0]["proj"], out[1]["pred"], out[1]["proj"]
    cos = torch.nn.CosineSimilarity(dim=1)
    loss = -(cos(p1, z2).mean() + cos(p2, z1).mean()) * 0.5
    loss.backward()
    res = {"p1": p1.detach().numpy(), "z1": z1.numpy(), "p2": p2.detach().numpy(), "z2": z2.numpy(),
           "loss": np.asarray(loss.item())}
    idx = np.random.default_rng(4).integers(0, 2 ** 31, size=64)
    for k, prm in net.named_parameters():
        gf = prm.grad.reshape(-1).numpy()
        res[f"grad_{k}_norm"] = np.asarray(np.linalg.norm(gf.astype(np.float64)))
        res[f"grad_{k}_sample"] = gf[idx % gf.size]
    res["sample_idx"] = idx
    res["feature_3d_running_var"] = net.feature_3d[1].running_var.numpy().copy()
    net2 = RS.TomoResClassifier(RS.BasicBlock, [2, 2, 2, 2], heads, 0)
    net2.load_state_dict(seeded_state_dict(net2, seed=319))
    net2.eval()
    with torch.no_grad():
        ft = net2.forward_test(x1[:1])              # b == 1: the permute branch of the reference
    res["test_proj_b1"] = ft["proj"].numpy()
    res["test_pred_b1"] = ft["pred"].numpy()
    save("simsiam_slices.npz", **res)
    path = os.path.join(HERE, "ckpt_keys.json")
    keys = json.load(open(path))
    keys["simsiam_18"] = {k: list(v.shape) for k, v in net.state_dict().items()}
    json.dump(keys, open(path, "w"), indent=0)


def gen_simsiam2d3d():
    """a3: TomoResClassifier2D3D (simsiam_model_2d3d.py:560-790), arch 'simsiam2d3d'."""
    from cet_pick.models.networks import simsiam_model_2d3d as R23
    heads = {"proj": 128, "pred": 128}
    net = R23.TomoResClassifier2D3D(R23.BasicBlock, [2, 2, 2, 2], heads, 128)
    net.load_state_dict(seeded_state_dict(net, seed=320))
    g = torch.Generator().manual_seed(9)
    xs = [torch.randn(4, 1, 28, 28, generator=g) for _ in range(4)]
    net.train()
    out = net(*xs)
    p1, z1, p2, z2 = out[0]["pred"], out[0]["proj"], out[1]["pred"], out[1]["proj"]
    cos = torch.nn.CosineSimilarity(dim=1)
    loss = -(cos(p1, z2).mean() + cos(p2, z1).mean()) * 0.5
    loss.backward()
    res = {"p1": p1.detach().numpy(), "z1": z1.numpy(), "p2": p2.detach().numpy(), "z2": z2.numpy(),
           "loss": np.asarray(loss.item())}
    for k, prm in net.named_parameters():
        res[f"grad_{k}_norm"] = np.asarray(np.linalg.norm(prm.grad.reshape(-1).numpy().astype(np.float64)))
    net.eval()
    with torch.no_grad():
        ft = net.forward_test(xs[0], xs[1])
    res["test_pred"] = ft["pred"].numpy()
    save("simsiam2d3d.npz", **res)
    path = os.path.join(HERE, "ckpt_keys.json")
    keys = json.load(open(path))
    keys["simsiam2d3d_18"] = {k: list(v.shape) for k, v in net.state_dict().items()}
    json.dump(keys, open(path, "w"), indent=0)


def gen_losses():
    """a23: _neg_loss, _pu_neg_loss, ConsistencyLoss, UnbiasedConLoss of cet_pick/models/loss.py with gradients."""
    from types import SimpleNamespace
    from cet_pick.models import loss as RL
    pred, gt, f, f_cr, lab, o1, o2 = losses_inputs()
    out = {}
    p = pred.clone().requires_grad_()
    l = RL._neg_loss(p, gt); l.backward()
    out["focal"], out["focal_grad"] = l.detach().numpy(), p.grad.numpy().copy()
    from cet_pick_amd.synthetic import confident_pred
    for tag, pr, tau in (("pu_0.05", pred, 0.05), ("pu_conf_0.6", confident_pred(gt), 0.6)):
        p = pr.clone().requires_grad_()
        l = RL._pu_neg_loss(p, gt, tau, 0, 1); l.backward()
        out[tag], out[tag + "_grad"] = l.detach().numpy(), p.grad.numpy().copy()
    a = o1.clone().requires_grad_()
    l = RL.ConsistencyLoss()(a, o2); l.backward()
    out["mse"], out["mse_grad"] = l.detach().numpy(), a.grad.numpy().copy()
    for thresh in (1.0, 0.4):
        opt = SimpleNamespace(thresh=thresh, device=torch.device("cpu"))
        fa, fb = f.clone().requires_grad_(), f_cr.clone().requires_grad_()
        pa, pb = o1.clone().requires_grad_(), o2.clone().requires_grad_()
        sup, unsup = RL.UnbiasedConLoss(0.07, 0.03)(lab, pa, pb, fa, fb, opt)
        (sup + 0.1 * unsup).backward()
        out[f"ucl_sup_{thresh}"], out[f"ucl_unsup_{thresh}"] = sup.detach().numpy(), unsup.detach().numpy()
        out[f"ucl_gf_{thresh}"], out[f"ucl_gfcr_{thresh}"] = fa.grad.numpy().copy(), fb.grad.numpy().copy()
        out[f"ucl_gp_{thresh}"], out[f"ucl_gpcr_{thresh}"] = pa.grad.numpy().copy(), pb.grad.numpy().copy()
    save("losses.npz", **out)


def save(name, **arrs):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrs)
    print("wrote", name, os.path.getsize(path) // 1024, "KiB")


def gen_decode():
    logits = make_logits((12, 28, 24), seed=11)
    x = torch.from_numpy(logits.copy())[None, None]
    hm = R_sigmoid(x)  # in place
    out = {"logits": logits, "sigmoid": hm[0, 0].numpy().copy()}
    for k in (3, 5):
        out[f"nms_3kk_{k}"] = R_decode._nms(hm, kernel=k)[0, 0].numpy()
        out[f"nms_xy_{k}"] = R_decode._nms_xy(hm, kernel=k)[0, 0].numpy()
        out[f"nms_z_{k}"] = R_decode._nms_z(hm, kernel=k)[0, 0].numpy()
        out[f"nms_kkk_{k}"] = R_image._nms(hm, kernel=k)[0, 0].numpy()
        out[f"decode_{k}"] = R_decode.tomo_decode(hm, kernel=k, K=50)[0].numpy()
        out[f"decode_fiber_{k}"] = R_decode.tomo_decode(hm, kernel=k, K=50, if_fiber=True)[0].numpy()
    s, z, y, xx, inds = R_decode._topk(R_decode._nms(hm, 3), K=40)
    out.update(topk_scores=s[0, 0].numpy(), topk_z=z[0].numpy(), topk_y=y[0].numpy(),
               topk_x=xx[0].numpy(), topk_inds=inds[0].numpy())
    # image.py variant (x = t % h) on H != W
    s2, z2, y2, x2, i2 = R_image._topk(R_image._nms(hm, 3), K=40)
    out.update(img_topk_z=z2[0].numpy(), img_topk_y=y2[0].numpy(), img_topk_x=x2[0].numpy())
    save("decode_small.npz", **out)


def gen_greedy():
    rng = np.random.default_rng(5)
    vol = rng.standard_normal((14, 20, 22)).astype(np.float32)
    out = {"vol": vol}
    for d, thr in ((6, 0.5), (14, 1.0), (4, -np.inf)):
        s, c = R_image.non_maximum_suppression_3d(vol, d, threshold=thr)
        out[f"scores_d{d}"] = s
        out[f"coords_d{d}"] = c
    s, c = R_decode.non_maximum_suppression_3d(vol.astype(np.float64), 6, scale=1.5, threshold=0.2)
    out["scores_d6_s15"] = s
    out["coords_d6_s15"] = c
    save("greedy_small.npz", **out)


def gen_dog():
    from scipy.ndimage import gaussian_filter
    shape = (36, 112, 104)
    vol, centres = make_tomo(shape, seed=317, margin_xy=40, margin_z=12)
    rec = vol.astype(np.float64)
    out = {"centres": centres, "shape": np.asarray(shape)}
    for sig in (2, 3, 5):
        g = gaussian_filter(rec, sig)
        out[f"gauss{sig}_z18"] = g[18, ::3].astype(np.float64)
        out[f"gauss{sig}_z0"] = g[0, ::3].astype(np.float64)
    for sigmas in ((2, 4), (3, 5), (2, 4, 6)):
        tag = "_".join(map(str, sigmas))
        s, c = R_image.get_potential_coords_pyramid(rec.copy(), sigmas=list(sigmas))
        out[f"scores_{tag}"] = s
        out[f"coords_{tag}"] = c
    # float32 input variant (what the GPU path consumes)
    s, c = R_image.get_potential_coords_pyramid(vol.copy(), sigmas=[3, 5])
    out["scores_3_5_f32in"] = s
    out["coords_3_5_f32in"] = c
    tz, ty, tx = R_image.get_potential_coords(rec.copy(), sigma1=2, sigma2=4, kernel=3, K=64)
    out.update(gpc_z=tz[0].numpy(), gpc_y=ty[0].numpy(), gpc_x=tx[0].numpy())
    save("dog_small.npz", **out)


HEADS = {"proj": 256, "pred": 256}


def _enc():
    enc = R_enc3d.TomoResClassifier3D(R_enc3d.BasicBlock, [2, 2, 2, 2], HEADS, 0)
    enc.load_state_dict(seeded_state_dict(enc, seed=317))
    return enc


def gen_enc3d():
    import contextlib
    import io
    enc = _enc()
    keys = {k: list(v.shape) for k, v in enc.state_dict().items()}
    g = torch.Generator().manual_seed(99)
    x = torch.randn(4, 1, 32, 32, 32, generator=g)
    acts = {}
    hooks = []
    for name in ("conv1", "bn1", "maxpool", "layer1", "layer2", "layer3", "feature_3d", "fc"):
        mod = getattr(enc, name)
        hooks.append(mod.register_forward_hook(
            lambda m, i, o, name=name: acts.__setitem__(name, o.detach().clone())))
    enc.train()
    with contextlib.redirect_stdout(io.StringIO()):
        out = enc(x)[0]["proj"]
    loss = (out * torch.linspace(-1, 1, 128)[None]).sum() + (out ** 2).sum() * 0.1
    loss.backward()
    res = {"proj_train": out.detach().numpy()}
    idx = np.random.default_rng(3).integers(0, 2 ** 31, size=256)
    for k, v in acts.items():
        f = v.reshape(-1).numpy()
        res[f"act_{k}_sample"] = f[idx % f.size]
        res[f"act_{k}_absmean"] = np.asarray(np.abs(f).mean())
    for k, p in enc.named_parameters():
        if p.grad is None:
            continue
        gf = p.grad.reshape(-1).numpy()
        res[f"grad_{k}_sample"] = gf[idx % gf.size]
        res[f"grad_{k}_norm"] = np.asarray(np.linalg.norm(gf.astype(np.float64)))
    res["bn1_running_mean"] = enc.bn1.running_mean.numpy().copy()
    res["bn1_running_var"] = enc.bn1.running_var.numpy().copy()
    res["proj7_running_var"] = enc.proj[7].running_var.numpy().copy()
    for h in hooks:
        h.remove()
    enc2 = _enc()
    enc2.eval()
    res["proj_eval"] = enc2.forward_test(x)["proj"].numpy()
    res["sample_idx"] = idx
    save("enc3d.npz", **res)
    path = os.path.join(HERE, "ckpt_keys.json")
    d = json.load(open(path)) if os.path.exists(path) else {}
    d.update({"moco3d_encoder": keys, "proj_is_pred": bool(enc.proj is enc.pred)})
    with open(path, "w") as f:
        json.dump(d, f, indent=1)


def gen_moco():
    """3 MoCo steps, SGD lr 0.05 (moco.py:101-146 restated around the hard-coded .cuda() at :141
    exactly as SURVEY.md §8c prescribes: the pieces are the reference's own methods)."""
    import contextlib
    import io
    torch.manual_seed(7)
    q, k = _enc(), _enc()
    moco = R_moco.MoCo(q, k, dim=128, r=64, m=0.99, T=0.1)
    g = torch.Generator().manual_seed(123)
    moco.queue.copy_(torch.nn.functional.normalize(torch.randn(128, 64, generator=g), dim=0))
    opt = torch.optim.SGD(moco.parameters(), lr=0.05)
    crit = torch.nn.CrossEntropyLoss()
    res = {"queue0": moco.queue.numpy().copy()}
    B = 8
    for step in range(3):
        im_q = torch.randn(B, 1, 32, 32, 32, generator=g)
        im_k = im_q.flip(4) + 0.1 * torch.randn(B, 1, 32, 32, 32, generator=g)
        with contextlib.redirect_stdout(io.StringIO()):
            qf = torch.nn.functional.normalize(moco.encoder_q(im_q)[0]["proj"], dim=1)
            with torch.no_grad():
                moco._momentum_update_key_encoder()
                kf = torch.nn.functional.normalize(moco.encoder_k(im_k)[0]["proj"], dim=1)
        l_pos = torch.einsum("nc,nc->n", [qf, kf]).unsqueeze(-1)
        l_neg = torch.einsum("nc,ck->nk", [qf, moco.queue.clone().detach()])
        logits = torch.cat([l_pos, l_neg], dim=1) / moco.T
        labels = torch.zeros(B, dtype=torch.long)
        moco._dequeue_and_enqueue(kf)
        loss = crit(logits, labels)
        opt.zero_grad()
        loss.backward()
        opt.step()
        res[f"logits_{step}"] = logits.detach().numpy()
        res[f"loss_{step}"] = np.asarray(loss.item())
        res[f"ptr_{step}"] = np.asarray(int(moco.queue_ptr))
    res["queue_final"] = moco.queue.numpy().copy()
    res["q_fc_weight"] = moco.encoder_q.fc.weight.detach().numpy().copy()
    res["k_fc_weight"] = moco.encoder_k.fc.weight.detach().numpy().copy()
    res["q_l1c1_sample"] = moco.encoder_q.layer1[0].conv1.weight.detach().reshape(-1)[::997].numpy().copy()
    res["k_l1c1_sample"] = moco.encoder_k.layer1[0].conv1.weight.detach().reshape(-1)[::997].numpy().copy()
    res["state_keys"] = np.asarray(list(moco.state_dict().keys()))
    save("moco_3steps.npz", **res)


def gen_moco_wc():
    """The same 3 MoCo steps as gen_moco at a WELL-CONDITIONED learning rate, so that gradients are comparable on every
    step.  Measured on the CPU oracle, fp32 against float64 of the same arithmetic (relative L2 of the
    stem / layer1 gradients at steps 0, 1, 2): lr 1e-3: 2e-5, 5e-3, 3e-1 (batch 8 and batch 32 alike: the step map
    amplifies a perturbation about 100x per step - 0.4 % weight change per step through the batch-statistics BatchNorms of
    the head and the un-normalised residual trunk); lr 1e-4: 2e-5, 1e-5, 2e-3; lr 1e-5: 2e-5, 1e-5, 1e-5.  Hence 1e-5.
    Per step: logits, loss, pointer, the norm of every parameter gradient and a sample of nine of them; at the end the
    weight DELTAS of three tensors (the update is ~1e-6 of a weight: deltas, not weights, show that SGD ran).
    -> moco_3steps_wc.npz"""
    import contextlib
    import io
    torch.manual_seed(7)
    q, k = _enc(), _enc()
    moco = R_moco.MoCo(q, k, dim=128, r=64, m=0.99, T=0.1)
    g = torch.Generator().manual_seed(123)
    moco.queue.copy_(torch.nn.functional.normalize(torch.randn(128, 64, generator=g), dim=0))
    opt = torch.optim.SGD(moco.parameters(), lr=1e-5)
    crit = torch.nn.CrossEntropyLoss()
    res = {"queue0": moco.queue.numpy().copy(), "lr": np.asarray(1e-5)}
    w0 = {n: prm.detach().clone() for n, prm in moco.encoder_q.named_parameters()}
    k0 = {n: prm.detach().clone() for n, prm in moco.encoder_k.named_parameters()}
    idx = np.random.default_rng(11).integers(0, 2 ** 31, size=32)
    res["sample_idx"] = idx
    sampled = ("conv1.weight", "layer1.0.conv1.weight", "layer2.0.conv1.weight", "layer2.0.downsample.0.weight",
               "layer3.0.conv1.weight", "layer3.1.conv2.weight", "feature_3d.0.weight", "fc.weight", "proj.0.weight")
    B = 8
    for step in range(3):
        im_q = torch.randn(B, 1, 32, 32, 32, generator=g)
        im_k = im_q.flip(4) + 0.1 * torch.randn(B, 1, 32, 32, 32, generator=g)
        with contextlib.redirect_stdout(io.StringIO()):
            qf = torch.nn.functional.normalize(moco.encoder_q(im_q)[0]["proj"], dim=1)
            with torch.no_grad():
                moco._momentum_update_key_encoder()
                kf = torch.nn.functional.normalize(moco.encoder_k(im_k)[0]["proj"], dim=1)
        l_pos = torch.einsum("nc,nc->n", [qf, kf]).unsqueeze(-1)
        l_neg = torch.einsum("nc,ck->nk", [qf, moco.queue.clone().detach()])
        logits = torch.cat([l_pos, l_neg], dim=1) / moco.T
        labels = torch.zeros(B, dtype=torch.long)
        moco._dequeue_and_enqueue(kf)
        loss = crit(logits, labels)
        opt.zero_grad()
        loss.backward()
        for n, prm in moco.encoder_q.named_parameters():
            if prm.grad is None or n.startswith("pred."):
                continue
            gf = prm.grad.reshape(-1).numpy()
            res[f"gnorm_{step}_{n}"] = np.asarray(np.linalg.norm(gf.astype(np.float64)))
            if n in sampled:
                res[f"gsample_{step}_{n}"] = gf[idx % gf.size].copy()
        opt.step()
        res[f"logits_{step}"] = logits.detach().numpy()
        res[f"loss_{step}"] = np.asarray(loss.item())
        res[f"ptr_{step}"] = np.asarray(int(moco.queue_ptr))
    res["queue_final"] = moco.queue.numpy().copy()
    qp, kp = dict(moco.encoder_q.named_parameters()), dict(moco.encoder_k.named_parameters())
    for n, stride in (("fc.weight", 7), ("layer1.0.conv1.weight", 997), ("layer3.0.downsample.0.weight", 101)):
        res[f"q_delta_{n}"] = (qp[n].detach().double() - w0[n].double()).reshape(-1)[::stride].numpy().copy()
        res[f"k_delta_{n}"] = (kp[n].detach().double() - k0[n].double()).reshape(-1)[::stride].numpy().copy()
    save("moco_3steps_wc.npz", **res)


def gen_simsiam2d():
    """a2 + a9: TomoResClassifier2D (simsiam_model_2d.py:617-819) two-view forward/backward under
    TomoSimSiamLoss's arithmetic (trains/tomo_simsiam_trainer.py:28-40; that module itself imports
    `progress`, so its six lines are evaluated here on the reference model's outputs)."""
    tvm = _stub("torchvision.models")
    _stub("torchvision.models.resnet", BasicBlock=object, Bottleneck=object, ResNet=object)
    sys.modules["torchvision"].models = tvm
    from cet_pick.models.networks import simsiam_model_2d as R2
    heads = {"proj": 128, "pred": 128}
    net = R2.TomoResClassifier2D(R2.BasicBlock, [2, 2, 2, 2], heads, 128)
    net.load_state_dict(seeded_state_dict(net, seed=318))
    keys = {k: list(v.shape) for k, v in net.state_dict().items()}
    g = torch.Generator().manual_seed(5)
    x1 = torch.randn(8, 1, 36, 36, generator=g)
    x2 = x1.flip(3) + 0.1 * torch.randn(8, 1, 36, 36, generator=g)
    net.train()
    out = net(x1, x2)
    p1, z1, p2, z2 = out[0]["pred"], out[0]["proj"], out[1]["pred"], out[1]["proj"]
    cos = torch.nn.CosineSimilarity(dim=1)
    loss = -(cos(p1, z2).mean() + cos(p2, z1).mean()) * 0.5
    ostd = torch.std(torch.nn.functional.normalize(p1.detach(), dim=1), 0).mean()
    loss.backward()
    res = {"p1": p1.detach().numpy(), "z1": z1.numpy(), "p2": p2.detach().numpy(), "z2": z2.numpy(),
           "loss": np.asarray(loss.item()), "output_std": np.asarray(ostd.item()),
           "z_requires_grad": np.asarray(int(z1.requires_grad))}
    idx = np.random.default_rng(4).integers(0, 2 ** 31, size=128)
    for k, prm in net.named_parameters():
        gf = prm.grad.reshape(-1).numpy()
        res[f"grad_{k}_norm"] = np.asarray(np.linalg.norm(gf.astype(np.float64)))
        res[f"grad_{k}_sample"] = gf[idx % gf.size]
    res["sample_idx"] = idx
    res["bn1_running_var"] = net.bn1.running_var.numpy().copy()
    res["nbt"] = np.asarray(int(net.bn1.num_batches_tracked))
    net2 = R2.TomoResClassifier2D(R2.BasicBlock, [2, 2, 2, 2], heads, 128)
    net2.load_state_dict(seeded_state_dict(net2, seed=318))
    net2.eval()
    ft = net2.forward_test(x1)
    res["test_proj"] = ft["proj"].numpy()
    res["test_pred"] = ft["pred"].detach().numpy()
    save("simsiam2d.npz", **res)
    path = os.path.join(HERE, "ckpt_keys.json")
    d = json.load(open(path))
    d["simsiam2d_encoder"] = keys
    json.dump(d, open(path, "w"), indent=1)


def gen_crops():
    """a13: the reference's own crop methods - TOMOPreProjAngleSelect3DVol.extract_subvols / extract_subvols_3d /
    extract_3d_tomo (datasets/tomo_pre_proj_angle_select_new3d_vol.py:110-138, called unbound: they only read crop sizes
    from `self`), the dataset mean / std of the stacked crops (:238-239, torch.mean / torch.std) and utils/loader.py `cutup`
    with the (2,4,4) stride of datasets/tomo_pre.py:104.  The module does cwd-relative imports (`from utils.image import
    ...`), so /root/reference/cet_pick goes on sys.path too; torchio (absent, unpinned in requirements.txt) is an inert
    stub: its transforms are never reached by these methods.  -> crops.npz"""
    from types import SimpleNamespace
    sys.path.insert(0, "/root/reference/cet_pick")
    _stub("torchio", Compose=None)
    from cet_pick.datasets import tomo_pre_proj_angle_select_new3d_vol as RD
    from cet_pick.utils import loader as R_loader
    cls = RD.TOMOPreProjAngleSelect3DVol
    vol, _ = make_tomo((20, 72, 80), seed=41, margin_xy=20, margin_z=4)
    rng = np.random.default_rng(9)
    n = 8
    coords = np.stack([rng.integers(16, 80 - 16, n), rng.integers(16, 72 - 16, n), rng.integers(3, 20 - 3, n)], 1).astype(np.int32)
    out = {"coords": coords}
    for size in ((3, 24, 24), (3, 16, 20), (5, 8, 12)):
        tag = "%d_%d_%d" % size
        sub2d = [cls.extract_subvols(None, vol, c, size) for c in coords]
        sub3d = [cls.extract_subvols_3d(None, vol, c, size) for c in coords]
        out["sub2d_" + tag] = torch.stack(sub2d).numpy()
        if size[1] < 24:
            out["sub3d_" + tag] = torch.stack(sub3d).numpy()
        st = torch.stack(sub2d)
        out["mean_" + tag] = np.float64(torch.mean(st).item())
        out["std_" + tag] = np.float64(torch.std(st).item())
    me = SimpleNamespace(crop_size_x=24, crop_size_y=16)
    out["tomo2d_24_16"] = torch.stack([cls.extract_3d_tomo(me, vol, c) for c in coords]).numpy()
    blks = R_loader.cutup(vol, (8, 64, 64), (2, 4, 4))
    out["cutup_shape"] = np.array(blks.shape)
    out["cutup_samples_idx"] = np.array([[0, 0, 0], [3, 1, 2], [6, 2, 4], [5, 0, 3]])
    out["cutup_samples"] = np.stack([blks[i, j, k] for i, j, k in out["cutup_samples_idx"]])
    save("crops.npz", **out)


def _stub_tree(*names):
    """inert, attribute-permissive stand-ins for absent packages whose names a reference module imports at load time
    but whose code the arithmetic under test never reaches (progress.bar, sknetwork.topology, pytorch_metric_learning)"""
    class _Any(types.ModuleType):
        def __getattr__(self, k):
            if k.startswith("__"):
                raise AttributeError(k)
            return None
    for name in names:
        parts = name.split(".")
        for i in range(1, len(parts) + 1):
            n = ".".join(parts[:i])
            if n not in sys.modules:
                m = _Any(n)
                m.__path__ = []
                sys.modules[n] = m


def gen_semi_loss():
    """a23 glue: the reference's own `TomoCRSemiLoss.forward` (trains/tomo_cr_semi_trainer.py:43-112), train phase with
    --contrastive, both `flip_prob` branches, values and gradients w.r.t. both views' heat-map logits and projections;
    plus the 'val' phase.  The module's load-time imports of progress / sknetwork / cv2 / utils.debugger are inert
    stubs: the loss arithmetic never touches them.  -> semi_loss.npz"""
    from types import SimpleNamespace
    _stub_tree("progress.bar", "sknetwork.topology", "pytorch_metric_learning")
    sys.modules["progress.bar"].Bar = object
    from cet_pick.trains import tomo_cr_semi_trainer as RT
    from cet_pick_amd.synthetic import semi_loss_inputs
    opt = SimpleNamespace(pn=False, ge=False, tau=0.1, temp=0.07, thresh=0.5, cr_weight=0.1, num_stacks=1,
                          contrastive=True, device=torch.device("cpu"))
    out = {}
    for fp in (0.2, 0.8):
        gt, hm, hm_cr, pj, pj_cr = semi_loss_inputs(fp)
        leaves = [t.clone().requires_grad_() for t in (hm, hm_cr, pj, pj_cr)]
        # `_sigmoid` works in place: hand it non-leaf tensors, as the network's outputs are
        o = [{"hm": leaves[0] * 1.0, "proj": leaves[2]}]
        o_cr = [{"hm": leaves[1] * 1.0, "proj": leaves[3]}]
        loss, stats = RT.TomoCRSemiLoss(opt)(o, {"hm": gt, "flip_prob": fp}, 1, "train", o_cr)
        loss.backward()
        tag = "%.1f" % fp
        for k in ("loss", "hm_loss", "cr_loss", "consis_loss"):
            out[f"{k}_{tag}"] = np.asarray(float(stats[k].detach()))
        for name, t in zip(("g_hm", "g_hm_cr", "g_proj", "g_proj_cr"), leaves):
            g = t.grad.numpy()
            out[f"{name}_{tag}"] = g.copy() if g.size < 4096 else g.reshape(-1)[::5].copy()     # projections: every 5th element
    gt, hm, _, _, _ = semi_loss_inputs(0.2)
    vloss, vstats = RT.TomoCRSemiLoss(opt)([{"hm": hm.clone(), "proj": None}], {"hm": gt}, 1, "val")
    out["val_loss"] = np.asarray(float(vloss))
    out["val_cr_loss"] = np.asarray(float(vstats["cr_loss"]))
    save("semi_loss.npz", **out)


def gen_moco_small():
    """f4: the reference's own `MoCoModel.forward` (trains/tomo_moco_small_trainer.py:24-161), symmetric and asymmetric,
    on the CPU.  The class hard-codes `.cuda()` on the shuffle index and the labels (:82,128): `torch.Tensor.cuda` is
    an identity for the duration of the call, nothing else is touched.  The encoders are the reference's
    TomoResClassifier3D behind a two-line adapter that returns the 'proj' tensor (MoCoModel expects an encoder that
    returns the embedding itself).  -> moco_small.npz"""
    import contextlib
    import io
    _stub_tree("progress.bar", "sknetwork.topology", "pytorch_metric_learning")
    sys.modules["progress.bar"].Bar = object
    from cet_pick.trains import tomo_moco_small_trainer as RM
    from cet_pick_amd.synthetic import moco_small_inputs

    class Proj(torch.nn.Module):
        def __init__(self, enc):
            super().__init__()
            self.enc = enc

        def forward(self, x):
            return self.enc(x)[0]["proj"]

    def enc330():
        enc = R_enc3d.TomoResClassifier3D(R_enc3d.BasicBlock, [2, 2, 2, 2], HEADS, 0)
        sd0 = seeded_state_dict(enc, seed=330)
        for kk in [k for k in sd0 if k.startswith("pred.")]:    # 'pred' re-registers the 'proj' module: one set of weights
            sd0["proj." + kk[5:]] = sd0[kk]
        enc.load_state_dict(sd0)
        return enc

    im1, im2, queue0 = moco_small_inputs()
    res = {}
    real_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        for tag, symmetric in (("sym", True), ("asym", False)):
            torch.manual_seed(5)                                  # the shuffle permutation
            model = RM.MoCoModel(Proj(enc330()), Proj(enc330()), None, dim=128, K=64, m=0.99, T=0.1, symmetric=symmetric)
            model.queue.copy_(queue0)
            model.train()
            with contextlib.redirect_stdout(io.StringIO()):
                loss, stats = model(im1, im2)
            loss.backward()
            res[f"loss_{tag}"] = np.asarray(float(loss))
            res[f"ptr_{tag}"] = np.asarray(int(model.queue_ptr))
            res[f"queue_{tag}"] = model.queue.numpy().copy()
            idx = np.random.default_rng(8).integers(0, 2 ** 31, size=64)
            for k, prm in model.encoder_q.enc.named_parameters():
                if prm.grad is None or k.startswith("pred."):
                    continue
                gf = prm.grad.reshape(-1).numpy()
                res[f"grad_{tag}_{k}_norm"] = np.asarray(np.linalg.norm(gf.astype(np.float64)))
                res[f"grad_{tag}_{k}_sample"] = gf[idx % gf.size]
            res["sample_idx"] = idx
            res[f"k_fc_weight_{tag}"] = model.encoder_k.enc.fc.weight.detach().reshape(-1)[::7].numpy().copy()
            res[f"k_l1c1_sample_{tag}"] = model.encoder_k.enc.layer1[0].conv1.weight.detach().reshape(-1)[::997].numpy().copy()
            res[f"k_bn1_running_mean_{tag}"] = model.encoder_k.enc.bn1.running_mean.numpy().copy()
    finally:
        torch.Tensor.cuda = real_cuda
    save("moco_small.npz", **res)


def gen_lr():
    class A:
        pass
    rows = []
    for cosine in (False, True):
        a = A()
        a.lr, a.cosine, a.lr_decay_rate, a.num_epochs, a.lr_step = 0.02, cosine, 0.1, 140, [90, 120]
        opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=a.lr)
        for ep in (1, 50, 90, 91, 120, 121, 140):
            R_utils.adjust_learning_rate(a, opt, ep)
            rows.append((float(cosine), ep, opt.param_groups[0]["lr"]))
    save("lr_sched.npz", rows=np.asarray(rows, dtype=np.float64))


if __name__ == "__main__":
    which = sys.argv[1:] or ["decode", "greedy", "dog", "enc3d", "moco", "lr", "simsiam2d", "loader", "unet", "losses", "simsiam", "simsiam2d3d", "crops", "semi_loss", "moco_small", "moco_wc"]
    for w in which:
        globals()["gen_" + w]()
