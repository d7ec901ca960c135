"""a2 + a9 on the GPU: 2-D convolutions through the implicit-GEMM kernel, the SimSiam 2-D encoder
two-view forward/backward and loss against the reference's golden vectors, and the simsiam trainer
entry points on BASELINE configs[0]-sized inputs (36x36 crops, batch 8)."""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("case", [(4, 36, 36, 64, 64, 3, 1, 1), (4, 36, 36, 64, 128, 3, 2, 1), (3, 18, 18, 64, 128, 1, 2, 0),
                                  (8, 36, 36, 1, 64, 3, 1, 1), (2, 9, 9, 256, 256, 3, 1, 1), (2, 13, 11, 32, 16, 3, 2, 1)])
def test_conv2d_fwd_dgrad_wgrad(case):
    from cet_pick_amd import hipops as H
    n, h, w, ci, co, k, s, p = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(n, ci, h, w, generator=g)
    wt = torch.randn(co, ci, k, k, generator=g) * (2.0 / (ci * k * k)) ** 0.5
    param = H.conv2d_weight_param(co, ci, k)
    with torch.no_grad():
        param.copy_(wt)
    param.data = param.data.cuda()
    assert H._phys_ok(param)
    xc = x.permute(0, 2, 3, 1).contiguous().cuda()
    y = H.conv_fwd(xc, param, k, s, p)
    xr, wr = x.clone().requires_grad_(True), wt.clone().requires_grad_(True)
    yr = F.conv2d(xr, wr, stride=s, padding=p)
    np.testing.assert_allclose(y.permute(0, 3, 1, 2).cpu().numpy(), yr.detach().numpy(), rtol=1e-4, atol=1e-4)
    dy = torch.randn(yr.shape, generator=g)
    yr.backward(dy)
    dyc = dy.permute(0, 2, 3, 1).contiguous().cuda()
    if ci != 1:
        dx = H.conv_dgrad(dyc, param, tuple(xc.shape), k, s, p)
        np.testing.assert_allclose(dx.permute(0, 3, 1, 2).cpu().numpy(), xr.grad.numpy(), rtol=1e-4, atol=2e-4)
    param.grad = None
    H.conv_wgrad_into(xc, dyc, param, k, s, p)
    sc = max(1.0, float(wr.grad.abs().max()))
    np.testing.assert_allclose(param.grad.cpu().numpy(), wr.grad.numpy(), rtol=1e-4, atol=2e-5 * sc)


def _oracle_grads(forward, sd0, inputs, dt):
    """Gradients of the SimSiam loss by the CPU oracle (oracle/train_ref.py) evaluated in dtype `dt`."""
    from oracle import train_ref as T
    sd = {k: (v.to(dt) if v.is_floating_point() else v.clone()).clone() for k, v in sd0.items()}
    names = [k for k in sd if k.endswith((".weight", ".bias"))]
    for n in names:
        sd[n].requires_grad_(True)
    p1, z1, p2, z2 = forward(sd, *[x.to(dt) for x in inputs], True)
    loss, _ = T.simsiam_loss(p1, z1, p2, z2)
    return dict(zip(names, torch.autograd.grad(loss, [sd[n] for n in names], allow_unused=True)))


def _arbitrated_grads(net, forward, sd0, inputs, g, norm_key="grad_%s_norm"):
    """Every parameter gradient of `net` against the oracle in float64, arbitrated by the oracle in fp32 (the GPU may sit as
    far from float64 as twice the CPU fp32 evaluation), and its norm against the reference's golden norm within the two
    measured fp32 errors.  Returns the worst GPU error."""
    from conftest import f32_equivalent
    g32 = _oracle_grads(forward, sd0, inputs, torch.float32)
    g64 = _oracle_grads(forward, sd0, inputs, torch.float64)
    gscale = max(float(v.norm()) for v in g64.values() if v is not None)
    worst = 0.0
    for n, p in net.named_parameters():
        if g64.get(n) is None:
            continue
        gf = p.grad.detach().cpu().contiguous()
        floor = 2e-5 * gscale / (float(g64[n].norm()) + 1e-30) + 2e-6
        e_g, e_c = f32_equivalent(gf.numpy(), g32[n].numpy(), g64[n].numpy(), floor=floor, what=n)
        worst = max(worst, e_g)
        ref = float(g[norm_key % n])                       # the reference's own fp32 run
        got = float(np.linalg.norm(gf.reshape(-1).numpy().astype(np.float64)))
        assert abs(got - ref) <= (e_g + e_c + floor + 1e-3) * ref + 2e-6, (n, got, ref)
    return worst


def _seeded_net():
    from cet_pick_amd.models.model import create_model
    from cet_pick_amd.synthetic import seeded_state_dict
    net = create_model("simsiam2d_18", {"proj": 128, "pred": 128}, 128)
    net.load_state_dict(seeded_state_dict(net, seed=318))
    return net.cuda()


def test_simsiam2d_matches_reference_golden(golden):
    from cet_pick_amd.trains.tomo_simsiam_trainer import TomoSimSiamLoss
    g = golden("simsiam2d.npz")
    net = _seeded_net()
    keys = json.load(open(os.path.join(HERE, "golden", "ckpt_keys.json")))["simsiam2d_encoder"]
    sd = net.state_dict()
    assert list(sd) == list(keys) and all(list(sd[k].shape) == keys[k] for k in keys)
    gen = torch.Generator().manual_seed(5)
    x1 = torch.randn(8, 1, 36, 36, generator=gen)
    x2 = x1.flip(3) + 0.1 * torch.randn(8, 1, 36, 36, generator=gen)
    net.train()
    out = net(x1.cuda(), x2.cuda())
    assert not out[0]["proj"].requires_grad and out[0]["pred"].requires_grad        # SimSiam stop-gradient
    for name, ref in (("p1", out[0]["pred"]), ("z1", out[0]["proj"]), ("p2", out[1]["pred"]), ("z2", out[1]["proj"])):
        np.testing.assert_allclose(ref.detach().cpu().numpy(), g[name], rtol=1e-3, atol=1e-3, err_msg=name)
    loss, stats = TomoSimSiamLoss(None)(out, None, 0)
    assert abs(float(loss.detach()) - float(g["loss"])) < 1e-4
    assert abs(float(stats["output_std"]) - float(g["output_std"])) < 1e-5
    loss.backward()
    # gradients: fp32 entries of the REFERENCE run are themselves far from a float64 evaluation (BatchNorm backward cancels
    # large terms in every block), so they are arbitrated by the oracle in float64 instead of held to a widened tolerance
    from oracle import train_ref as T
    from cet_pick_amd.synthetic import seeded_state_dict
    sd0 = seeded_state_dict(net, seed=318)
    _arbitrated_grads(net, T.simsiam_forward, sd0, (x1, x2), g)
    idx = g["sample_idx"]
    g32 = _oracle_grads(T.simsiam_forward, sd0, (x1, x2), torch.float32)
    for n, p in net.named_parameters():                      # the oracle's fp32 samples ARE the reference's (CPU test)
        np.testing.assert_allclose(g32[n].reshape(-1).numpy()[idx % g32[n].numel()], g[f"grad_{n}_sample"], rtol=0,
                                   atol=2e-3 * float(np.abs(g[f"grad_{n}_sample"]).max()) + 2e-6, err_msg=n)
    np.testing.assert_allclose(net.bn1.running_var.cpu().numpy(), g["bn1_running_var"], rtol=1e-4, atol=1e-5)
    assert int(net.bn1.num_batches_tracked) == int(g["nbt"])          # two views -> two updates
    net2 = _seeded_net()
    net2.eval()
    ft = net2.forward_test(x1.cuda())
    np.testing.assert_allclose(ft["proj"].cpu().numpy(), g["test_proj"], rtol=1e-3, atol=1e-3)
    np.testing.assert_allclose(ft["pred"].detach().cpu().numpy(), g["test_pred"], rtol=1e-3, atol=1e-3)


def test_simsiam_trainer_entry_points(tmp_path, monkeypatch):
    """configs[0]-style plumbing on the GPU: opts -> create_model -> train_factory['simsiam3d'] ->
    trainer.train with an SGD optimizer over the strided kernel-layout parameters."""
    from cet_pick_amd.opts import opts
    from cet_pick_amd.models.model import create_model, save_model, load_model
    from cet_pick_amd.trains.train_factory import train_factory
    monkeypatch.chdir(tmp_path)
    opt = opts().parse(["simsiam3d", "--arch", "simsiam2d_18", "--batch_size", "8", "--lr", "0.05", "--debug", "0",
                        "--bbox", "36"])
    opt.heads = {"proj": opt.head_conv, "pred": opt.head_conv}
    assert opt.head_conv == 128
    model = create_model(opt.arch, opt.heads, opt.head_conv, local_path=opt.pretrained_model)
    optimizer = torch.optim.SGD(filter(lambda p: p.requires_grad, model.parameters()), opt.lr)
    trainer = train_factory[opt.task](opt, model, optimizer)
    trainer.set_device(opt.gpus, opt.chunk_sizes, torch.device("cuda"))
    gen = torch.Generator().manual_seed(1)
    base = torch.randn(32, 1, 36, 36, generator=gen)
    loader = [{"input": base[i:i + 8], "input_aug": base[i:i + 8].flip(3) + 0.05 * torch.randn(8, 1, 36, 36, generator=gen)}
              for i in range(0, 32, 8)]
    first, _ = trainer.train(1, loader)
    for ep in range(2, 5):
        last, _ = trainer.train(ep, loader)
    assert set(first) == {"loss", "cosine_loss", "output_std", "time"}
    assert np.isfinite(last["loss"]) and last["loss"] < first["loss"]        # cosine loss decreases
    path = str(tmp_path / "m.pth")
    save_model(path, 4, model, optimizer)
    m2 = load_model(create_model(opt.arch, opt.heads, opt.head_conv), path)
    assert torch.equal(m2.fc.weight.cpu(), model.fc.weight.detach().cpu())


def test_simsiam_slicewise_encoder_matches_reference_golden(golden):
    """row a3 (arch 'simsiam'): per-slice 2-D trunk + Conv3d/BN3d head, two-view forward / backward and eval, against
    outputs of the reference's TomoResClassifier (simsiam_model.py:159-440)."""
    from cet_pick_amd.models.model import create_model
    from cet_pick_amd.synthetic import seeded_state_dict
    from cet_pick_amd.trains.tomo_simsiam_trainer import TomoSimSiamLoss
    g = golden("simsiam_slices.npz")
    heads = {"proj": 256, "pred": 256}

    def make():
        net = create_model("simsiam_18", heads, 0)
        net.load_state_dict(seeded_state_dict(net, seed=319))
        return net.cuda()
    net = make()
    keys = json.load(open(os.path.join(HERE, "golden", "ckpt_keys.json")))["simsiam_18"]
    sd = net.state_dict()
    assert list(sd) == list(keys) and all(list(sd[k].shape) == keys[k] for k in keys)
    gen = torch.Generator().manual_seed(6)
    x1 = torch.randn(4, 5, 40, 40, generator=gen)
    x2 = x1.flip(3) + 0.1 * torch.randn(4, 5, 40, 40, generator=gen)
    net.train()
    out = net(x1.cuda(), x2.cuda())
    for name, ref in (("p1", out[0]["pred"]), ("z1", out[0]["proj"]), ("p2", out[1]["pred"]), ("z2", out[1]["proj"])):
        np.testing.assert_allclose(ref.detach().cpu().numpy(), g[name], rtol=2e-3, atol=2e-3, err_msg=name)
    loss, _ = TomoSimSiamLoss(None)(out, None, 0)
    assert abs(float(loss.detach()) - float(g["loss"])) < 2e-4
    loss.backward()
    from oracle import train_ref as T
    from cet_pick_amd.synthetic import seeded_state_dict
    _arbitrated_grads(net, T.simsiam_slice_forward, seeded_state_dict(net, seed=319), (x1, x2), g)
    np.testing.assert_allclose(net.feature_3d[1].running_var.cpu().numpy(), g["feature_3d_running_var"], rtol=1e-3, atol=1e-5)
    net2 = make().eval()
    with torch.no_grad():
        ft = net2.forward_test(x1[:1].cuda())
    np.testing.assert_allclose(ft["proj"].cpu().numpy(), g["test_proj_b1"], rtol=2e-3, atol=2e-3)
    np.testing.assert_allclose(ft["pred"].cpu().numpy(), g["test_pred_b1"], rtol=2e-3, atol=2e-3)


def test_simsiam2d3d_encoder_matches_reference_golden(golden):
    """row a3 (arch 'simsiam2d3d'): tilt + tomogram patches through the shared trunk, against the reference."""
    from cet_pick_amd.models.model import create_model
    from cet_pick_amd.synthetic import seeded_state_dict
    from cet_pick_amd.trains.tomo_simsiam_trainer import TomoSimSiamLoss
    g = golden("simsiam2d3d.npz")
    net = create_model("simsiam2d3d_18", {"proj": 128, "pred": 128}, 128)
    net.load_state_dict(seeded_state_dict(net, seed=320))
    net = net.cuda().train()
    gen = torch.Generator().manual_seed(9)
    xs = [torch.randn(4, 1, 28, 28, generator=gen).cuda() for _ in range(4)]
    out = net(*xs)
    for name, ref in (("p1", out[0]["pred"]), ("z1", out[0]["proj"]), ("p2", out[1]["pred"]), ("z2", out[1]["proj"])):
        np.testing.assert_allclose(ref.detach().cpu().numpy(), g[name], rtol=2e-3, atol=2e-3, err_msg=name)
    loss, _ = TomoSimSiamLoss(None)(out, None, 0)
    assert abs(float(loss.detach()) - float(g["loss"])) < 2e-4
    loss.backward()
    from oracle import train_ref as T
    _arbitrated_grads(net, T.simsiam2d3d_forward, seeded_state_dict(net, seed=320), tuple(x.cpu() for x in xs), g)
    net.eval()
    with torch.no_grad():
        ft = net.forward_test(xs[0], xs[1])
    np.testing.assert_allclose(ft["pred"].cpu().numpy(), g["test_pred"], rtol=2e-3, atol=2e-3)


@pytest.mark.parametrize("case", [(5, 36, 36, 64), (3, 40, 36, 64), (7, 18, 18, 128), (9, 9, 9, 256), (37, 9, 9, 256),
                                  # the generic instance (run-time geometry): the default --bbox 32 and others, H != W, a one-row plane
                                  (4, 32, 32, 64), (6, 16, 16, 128), (10, 8, 8, 256), (3, 24, 20, 64), (2, 48, 48, 64), (2, 64, 64, 128),
                                  (5, 12, 12, 128), (9, 6, 6, 256), (3, 20, 36, 64), (7, 1, 9, 64)])
def test_conv2d_direct_plane_kernel_matches_the_implicit_gemm_and_float64(case, monkeypatch):
    """conv_p2d.hip (round 6): the 3 x 3 / stride-1 layers of the 2-D encoder, forward and data gradient (with the residual /
    mask / ReLU epilogues), against float64 and against the implicit GEMM - batch sizes whose flat voxel run ends inside a
    128-voxel tile, planes that straddle tiles, a plane taller than wide."""
    from cet_pick_amd import hipops as H
    n, h, w, c = case
    g = torch.Generator().manual_seed(sum(case))
    tol = 2e-6 * max(1.0, (c / 64) ** 0.5)      # f32 accumulation over K = 9 c terms: rounding noise grows with sqrt(K)
    x = torch.randn(n, c, h, w, generator=g)
    wt = torch.randn(c, c, 3, 3, generator=g) * (2.0 / (c * 9)) ** 0.5
    param = H.conv2d_weight_param(c, c, 3)
    with torch.no_grad():
        param.copy_(wt)
    param.data = param.data.cuda()
    xc = x.permute(0, 2, 3, 1).contiguous().cuda()
    assert H.p2d_usable(tuple(xc.shape), c, c, (1, 3, 3), 1, (0, 1, 1))
    from cet_pick_amd import _lib as L
    kind = int(L.lib().mi_conv2d_p2d_usable(n, h, w, c))
    assert kind == (1 if (w, c) in ((36, 64), (18, 128), (9, 256)) and h >= w else 2)
    # the weight gradient has its kernel wherever the X window of a 64-voxel K-block has <= 192 rows (W <= ~40)
    assert (int(L.lib().mi_conv2d_p2d_wgrad_workspace_bytes(n, h, w, c)) > 0) == (w <= 40), (n, h, w, c)
    res = torch.randn(n, h, w, c, generator=g).cuda()
    y = H.conv_fwd(xc, param, 3, 1, 1)
    y_res = H.conv_fwd(xc, param, 3, 1, 1, res=res, relu=True)
    y64 = F.conv2d(x.double(), wt.double(), padding=1).permute(0, 2, 3, 1)
    sc = float(y64.abs().max())
    assert float((y.cpu().double() - y64).abs().max()) <= tol * sc
    assert float((y_res.cpu().double() - torch.relu(y64 + res.cpu().double())).abs().max()) <= tol * sc
    dy = torch.randn(n, h, w, c, generator=g).cuda()
    mask = torch.randn(n, h, w, c, generator=g).cuda()
    dx = H.conv_dgrad(dy, param, tuple(xc.shape), 3, 1, 1)
    dx_m = H.conv_dgrad(dy, param, tuple(xc.shape), 3, 1, 1, res=res, mask=mask)
    dx64 = F.conv_transpose2d(dy.cpu().double().permute(0, 3, 1, 2), wt.double(), padding=1).permute(0, 2, 3, 1)
    sc = float(dx64.abs().max())
    assert float((dx.cpu().double() - dx64).abs().max()) <= tol * sc
    want = (dx64 + res.cpu().double()) * (mask.cpu() > 0)
    assert float((dx_m.cpu().double() - want).abs().max()) <= tol * sc
    # weight gradient (p2d_wgrad_kernel) against float64, and against the implicit GEMM below
    xr, wr = x.double().requires_grad_(True), wt.double().requires_grad_(True)
    (gw64,) = torch.autograd.grad(F.conv2d(xr, wr, padding=1), wr, dy.cpu().double().permute(0, 3, 1, 2))
    param.grad = None
    H.conv_wgrad_into(xc, dy, param, 3, 1, 1)
    gw = param.grad.detach().clone()
    wtol = 2e-6 * max(1.0, (n * h * w / 512) ** 0.5)       # f32 accumulation over n h w voxels
    assert float((gw.cpu().double() - gw64).abs().max()) <= wtol * float(gw64.abs().max())
    param.grad = None
    # the weights change (a torch op bumps the version): the cached images follow
    with torch.no_grad():
        param.mul_(0.5)
    assert float((H.conv_fwd(xc, param, 3, 1, 1).cpu().double() - 0.5 * y64).abs().max()) <= tol * float(y64.abs().max())
    # same products on the implicit GEMM: rounding-order noise only
    monkeypatch.setenv("MI_NO_P2D", "1")
    assert not H.p2d_usable(tuple(xc.shape), c, c, (1, 3, 3), 1, (0, 1, 1))
    y_ig = H.conv_fwd(xc, param, 3, 1, 1)
    assert float((y_ig.cpu().double() - 0.5 * y64).abs().max()) <= tol * float(y64.abs().max())
    monkeypatch.setenv("MI_NO_P2D_WGRAD", "1")
    param.grad = None
    H.conv_wgrad_into(xc, dy, param, 3, 1, 1)
    assert float((param.grad.cpu().double() - gw64).abs().max()) <= wtol * float(gw64.abs().max())


def _simsiam_trainer(seed, hipgraph, engine=True, monkeypatch=None, lr=0.05):
    from types import SimpleNamespace
    from cet_pick_amd.models.model import create_model
    from cet_pick_amd.synthetic import seeded_state_dict
    from cet_pick_amd.trains.train_factory import train_factory
    net = create_model("simsiam2d_18", {"proj": 128, "pred": 128}, 128)
    net.load_state_dict(seeded_state_dict(net, seed=seed))
    opt = SimpleNamespace(task="simsiam3d", num_iters=-1, print_iter=0, hide_data_time=True, exp_id="t", lr=lr, hipgraph=hipgraph)
    if not engine:
        monkeypatch.setenv("CETPICK_SIMSIAM_ENGINE", "0")
    tr = train_factory["simsiam3d"](opt, net, torch.optim.SGD(net.parameters(), lr=lr))
    tr.set_device([0], None, "cuda")
    if not engine:
        monkeypatch.delenv("CETPICK_SIMSIAM_ENGINE")
    return net, tr


def test_simsiam_engine_step_equals_plain_sequence(monkeypatch):
    """SimSiamStepEngine (round 6: flat arenas, second gradient arena instead of a grad.add_ per parameter, fused SGD kernel, cached
    weight images, device meters, hipGraph replay from the third call on) against the plain sequence `model(x1, x2); loss; zero_grad;
    backward; sgd_step_` on the same seeded weights and batches: the parameters after every one of five steps, the losses and the
    meters - BIT FOR BIT (same kernels on the same operands; g1 + g2 is one rounding either way)."""
    from cet_pick_amd import hipops as H
    net_e, tr_e = _simsiam_trainer(321, hipgraph=True)
    net_p, tr_p = _simsiam_trainer(321, hipgraph=False, engine=False, monkeypatch=monkeypatch)
    assert tr_e.engine is not None and tr_e.engine.use_graph and tr_p.engine is None
    arena_p = H.ParamArena(net_p)
    g = torch.Generator().manual_seed(4)
    losses = []
    for it in range(5):
        x1 = torch.randn(16, 1, 36, 36, generator=g).cuda()
        x2 = (x1.flip(-1) + 0.1 * torch.randn(16, 1, 36, 36, generator=g).cuda()).contiguous()
        le = tr_e.train_step(x1, x2)
        # plain sequence
        arena_p.zero_grad()
        tr_p.model_with_loss.train()
        _, lp, stats = tr_p.model_with_loss({"input": x1, "input_aug": x2}, 0, "train")
        lp.backward()
        H.sgd_step_(arena_p.flat, arena_p.flat_grad, 0.05)
        assert torch.equal(le, lp.detach()), (it, float(le), float(lp))
        assert torch.equal(tr_e.engine.arena.flat, arena_p.flat), it
        for (n1, b1), (n2, b2) in zip(net_e.named_buffers(), net_p.named_buffers()):
            assert torch.equal(b1, b2), (it, n1)
        losses.append(float(lp))
    assert tr_e.engine._graph is not None                      # steps 3.. were graph replays
    nodes = tr_e.engine.node_counts()
    assert nodes["memcpy"] == 0 and nodes["kernel"] > 100
    sums = tr_e.engine.take_stat_sums()
    assert abs(sums["loss"] - sum(losses)) <= 1e-5 * abs(sum(losses)) and sums["cosine_loss"] == sums["loss"] and sums["output_std"] > 0
    # a write through torch ops between two steps (a checkpoint load): the engine re-cuts its weight images by itself
    with torch.no_grad():
        tr_e.engine.arena.flat.mul_(0.999)
        arena_p.flat.mul_(0.999)
    le = tr_e.train_step(x1, x2)
    arena_p.zero_grad()
    _, lp, _ = tr_p.model_with_loss({"input": x1, "input_aug": x2}, 0, "train")
    lp.backward()
    H.sgd_step_(arena_p.flat, arena_p.flat_grad, 0.05)
    assert torch.equal(le, lp.detach()) and torch.equal(tr_e.engine.arena.flat, arena_p.flat)
    tr_e.close()


@pytest.mark.parametrize("n,hw,cin,planes,stride", [(24, 18, 128, 128, 1), (6, 36, 64, 64, 1), (8, 36, 64, 128, 2), (40, 8, 64, 64, 1)])
def test_block_residual_gradient_slots_equal_autograd_sum(n, hw, cin, planes, stride, monkeypatch):
    """hipops.GradSlot (round 6): the gradient a BasicBlock's input receives through the shortcut - from bn2's backward (identity shortcut;
    mi_bn_bwd_apply_res writes the masked gradient on the way) or from the shortcut convolution's data gradient - is added in conv1's
    data-gradient epilogue instead of by autograd's sum.  Same single rounding: the input gradient and every parameter gradient are
    BIT-EQUAL to the plain form (CETPICK_FUSE_RES_GRAD=0); both match torch in float64.  The last case takes the small-BatchNorm path
    (one-launch backward, no slot for the mask), the others the statistics + apply passes."""
    import copy
    import torch.nn as nn
    from cet_pick_amd import hipops as H
    from cet_pick_amd.models.networks.simsiam_model_2d import BasicBlock
    g = torch.Generator().manual_seed(n * 1000 + hw)
    ds = None
    if stride != 1 or cin != planes:
        ds = nn.Sequential(H.HipConv2d(cin, planes, 1, stride=stride, pad=0))
    blk0 = BasicBlock(cin, planes, stride, ds).cuda()
    with torch.no_grad():
        for p in blk0.parameters():
            if p.dim() == 1:
                p.copy_((torch.rand(p.shape, generator=g) + 0.5).cuda())
    x0 = torch.randn(n, hw, hw, cin, generator=g)
    ho = hw // stride
    dy = torch.randn(n, ho, ho, planes, generator=g)

    def run(fused):
        monkeypatch.setattr(H, "FUSE_RES_GRAD", fused)
        blk = copy.deepcopy(blk0)
        blk.train()
        x = x0.cuda().requires_grad_(True)
        xin = x * 1.0                                        # (a non-leaf input, as inside the network)
        y = blk(xin)
        y.backward(dy.cuda())
        return [y.detach(), x.grad] + [p.grad for p in blk.parameters()]

    a, b = run(True), run(False)
    assert len(a) == len(b)
    for u, v in zip(a, b):
        assert torch.equal(u, v)
    # float64 reference (NCHW torch modules with the same parameters)
    import torch.nn.functional as F
    sd = {k: v.detach().cpu().double() for k, v in blk0.state_dict().items()}
    xr = x0.double().permute(0, 3, 1, 2).requires_grad_(True)
    def bn(t, pre):
        return F.batch_norm(t, None, None, sd[pre + ".weight"], sd[pre + ".bias"], True, 0.1, 1e-5)
    h = torch.relu(bn(F.conv2d(xr, sd["conv1.weight"], stride=stride, padding=1), "bn1"))
    h = bn(F.conv2d(h, sd["conv2.weight"], padding=1), "bn2")
    res = xr if ds is None else F.conv2d(xr, sd["downsample.0.weight"], stride=stride)
    yr = torch.relu(h + res)
    yr.backward(dy.double().permute(0, 3, 1, 2))
    got_y, got_dx = a[0].cpu().double(), a[1].cpu().double()
    assert float((got_y - yr.detach().permute(0, 2, 3, 1)).abs().max()) <= 2e-5 * float(yr.detach().abs().max())
    # (a unit within fp32 rounding of zero may take the other branch of a ReLU than float64 does - its gradient then differs by O(1) in
    # a few elements: the bulk is held tightly, the whole in norm)
    err = (got_dx - xr.grad.detach().permute(0, 2, 3, 1)).abs()
    scale = float(xr.grad.abs().max())
    assert float((err <= 5e-5 * scale).double().mean()) >= 0.99          # (a flipped unit reaches a 5 x 5 neighbourhood of dx through two convolutions)
    assert float(err.norm()) <= 1e-2 * float(xr.grad.norm())
