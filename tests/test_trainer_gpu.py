"""The reference's entry points for the train path, end to end on the GPU: flags -> create_model ->
MoCo -> train_factory -> trainer.train(epoch, loader) -> save_model / load_model."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_moco_main_two_epochs_and_resume(tmp_path, monkeypatch):
    from cet_pick_amd.opts import opts
    from cet_pick_amd import moco_main
    from cet_pick_amd.models.model import create_model, load_model
    from cet_pick_amd.models.moco import MoCo
    monkeypatch.chdir(tmp_path)
    args = ["moco", "--arch", "moco3d_18", "--batch_size", "16", "--num_epochs", "2", "--lr", "0.01", "--lr_step", "1",
            "--exp_id", "t", "--debug", "0", "--print_iter", "4", "--val_intervals", "2"]
    moco_main.main(opts().parse(args))
    save_dir = os.path.join(str(tmp_path), "exp", "moco", "t")
    lines = open(os.path.join(save_dir, "log.txt")).read().strip().split("\n")
    assert len(lines) == 2 and lines[0].startswith("epoch: 1 |loss ") and "infoNCE" in lines[0] and "time" in lines[0]
    losses = [float(l.split("|")[1].split()[1]) for l in lines]
    assert all(np.isfinite(losses)) and losses[1] < losses[0] + 0.5
    assert os.path.exists(os.path.join(save_dir, "model_last_contrastive.pth"))      # epoch 1
    ck = torch.load(os.path.join(save_dir, "model_last.pth"))                          # epoch 2 (val interval)
    assert set(ck) == {"epoch", "state_dict", "optimizer"} and ck["epoch"] == 2
    assert len(ck["state_dict"]) == 122 and "queue" in ck["state_dict"] and "encoder_k.fc.weight" in ck["state_dict"]
    assert os.path.exists(os.path.join(save_dir, "model_1.pth"))
    heads = {"proj": 256, "pred": 256}
    m = MoCo(create_model("moco3d_18", heads, 0), create_model("moco3d_18", heads, 0), dim=128)
    m = load_model(m, os.path.join(save_dir, "model_last.pth"))
    assert torch.equal(m.queue, ck["state_dict"]["queue"]) and int(m.queue_ptr) == int(ck["state_dict"]["queue_ptr"])
    # the key encoder trails the query encoder (EMA), neither is the initial state
    assert not torch.equal(m.encoder_q.fc.weight, m.encoder_k.fc.weight)
    # resume continues from the stored epoch with the decayed lr
    moco_main.main(opts().parse(args[:5] + ["--num_epochs", "3", "--lr", "0.01", "--lr_step", "1", "--exp_id", "t",
                                            "--debug", "0", "--resume"]))
    assert len(open(os.path.join(save_dir, "log.txt")).read().strip().split("\n")) == 3


def test_moco_main_hipgraph_follows_lr_schedule_and_equals_eager(tmp_path, monkeypatch):
    """The CLI's default (--hipgraph for task moco) replays the step from a hipGraph: the per-epoch learning rate
    reaches it through the engine's device scalar, and two epochs end in the weights of the eager run bit for bit."""
    from cet_pick_amd.opts import opts
    from cet_pick_amd import moco_main
    monkeypatch.chdir(tmp_path)
    base = ["moco", "--arch", "moco3d_18", "--batch_size", "16", "--num_epochs", "2", "--lr", "0.02", "--lr_step", "1",
            "--debug", "0", "--val_intervals", "2"]
    o = opts().parse(base + ["--exp_id", "g"])
    assert o.hipgraph is True and opts().parse(["semi"]).hipgraph is False
    seen = []
    from cet_pick_amd.trains.moco_engine import MocoStepEngine
    orig = MocoStepEngine.set_lr

    def spy(self, lr):
        orig(self, lr)
        seen.append((lr, float(self.lr_dev.item()), self.use_graph))
    monkeypatch.setattr(MocoStepEngine, "set_lr", spy)
    moco_main.main(o)
    assert [round(a, 6) for a, _, _ in seen] == [0.02, 0.002] and all(abs(a - b) < 1e-9 for a, b, _ in seen)
    assert all(g for _, _, g in seen)
    moco_main.main(opts().parse(base + ["--exp_id", "e", "--no_hipgraph"]))
    assert seen[-1][2] is False
    a = torch.load(os.path.join(str(tmp_path), "exp", "moco", "g", "model_last.pth"))["state_dict"]
    b = torch.load(os.path.join(str(tmp_path), "exp", "moco", "e", "model_last.pth"))["state_dict"]
    assert set(a) == set(b)
    for k in a:
        assert torch.equal(a[k], b[k]), k


def test_engine_short_batch_runs_eagerly():
    """A batch whose shape differs from the captured one never reaches the graph's static buffers."""
    from cet_pick_amd.models.networks.moco_encoder_3d import get_moco_net_small_3d
    from cet_pick_amd.models.moco import MoCo
    from cet_pick_amd.trains.moco_engine import MocoStepEngine
    torch.manual_seed(3)
    heads = {"proj": 256, "pred": 256}
    moco = MoCo(get_moco_net_small_3d(18, heads, 0), get_moco_net_small_3d(18, heads, 0), dim=128, r=64).cuda().train()
    eng = MocoStepEngine(moco, lr=1e-3, use_graph=True)
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn(8, 1, 32, 32, 32, device="cuda", generator=g)
    for _ in range(4):
        eng.step(x, x.flip(4))
    assert eng._graph is not None
    ptr = int(moco.queue_ptr)
    l = eng.step(x[:4], x[:4].flip(4))                       # short batch: eager, queue advances by 4
    assert np.isfinite(float(l)) and int(moco.queue_ptr) == (ptr + 4) % 64
    l = eng.step(x, x.flip(4))                               # and the graph still replays afterwards
    assert np.isfinite(float(l)) and int(moco.queue_ptr) == (ptr + 12) % 64
    eng.close()
    assert eng._graph is None


def test_semi_detector_trainer_two_steps_vs_oracle(tmp_path):
    """unet_4 detector training (task 'semi', --contrastive) through TomoCRSemiTrainer; the first
    step's loss terms and the updated weights are checked against torch autograd on the CPU oracles."""
    from types import SimpleNamespace
    import numpy as np
    from oracle import loss_ref as OL, unet_ref as OU
    from cet_pick_amd.models.model import create_model
    from cet_pick_amd.synthetic import seeded_state_dict
    from cet_pick_amd.trains.train_factory import train_factory
    heads = {"hm": 1, "proj": 32}
    opt = SimpleNamespace(task="semi", arch="unet_4", pn=False, ge=False, tau=0.1, temp=0.07, thresh=0.5, cr_weight=0.1,
                          num_stacks=1, contrastive=True, device=torch.device("cuda"), num_iters=-1, print_iter=0,
                          hide_data_time=True, exp_id="t", lr=1e-3, hipgraph=False)
    model = create_model(opt.arch, heads, 32)
    sd0 = seeded_state_dict(model, seed=323)
    for k in ("hm.weight", "proj.weight"):
        sd0[k] = sd0[k] * 0.3
    model.load_state_dict(sd0)
    optim = torch.optim.SGD(model.parameters(), lr=opt.lr)
    trainer = train_factory["semi"](opt, model, optim)
    trainer.set_device([0], None, "cuda")
    g = torch.Generator().manual_seed(2)
    b, d, h, w = 2, 4, 48, 48
    x = torch.randn(b, d, h, w, generator=g)
    x_aug = x.flip(-1) + 0.05 * torch.randn(b, d, h, w, generator=g)
    gt = torch.full((b, 1, d, h // 2, w // 2), -1.0)
    r = torch.rand(gt.shape, generator=g)
    gt[r < 0.3] = 0.0
    gt[(r >= 0.3) & (r < 0.4)] = 0.6
    gt[r > 0.96] = 1.0
    batch = {"input": x, "input_aug": x_aug, "hm": gt, "flip_prob": 0.2, "meta": {}}
    # reference step on the CPU, in fp32 and - the arbiter - in float64
    def cpu_step(dt):
        rsd = {k: (v.to(dt) if v.is_floating_point() else v.clone()).clone().requires_grad_(
            v.is_floating_point() and not k.endswith(("running_mean", "running_var"))) for k, v in sd0.items()}
        o1 = OU.tomo_conv_unet_forward(rsd, x.to(dt), 4, heads, training=True)
        o2 = OU.tomo_conv_unet_forward(rsd, x_aug.to(dt), 4, heads, training=True)
        r = OL.tomo_cr_semi_loss(o1["hm"], o2["hm"], o1["proj"], o2["proj"], gt.to(dt), 0.2, opt.tau, opt.temp, opt.thresh,
                                 opt.cr_weight)
        r[0].backward()
        return rsd, r
    ref_sd, ref = cpu_step(torch.float32)
    ref_sd64, ref64 = cpu_step(torch.float64)
    stats, _ = trainer.train(1, [dict(batch)])
    assert set(stats) == {"loss", "hm_loss", "cr_loss", "consis_loss", "time"}
    for k, v, v64 in zip(("loss", "hm_loss", "cr_loss", "consis_loss"), ref, ref64):
        assert abs(stats[k] - v64.item()) <= 2 * abs(v.item() - v64.item()) + 1e-4 * abs(v64.item()) + 1e-7, k
    from conftest import f32_equivalent
    for name, prm in model.named_parameters():
        if name.endswith("upconv.bias"):
            continue
        # the step's gradient (still in .grad after the step) against the float64 step: the GPU may be as far from it
        # as twice torch's own fp32 step.  (The weight DELTA is no measure of the gradient: lr * grad is a few ulps of
        # the weight itself - which is what a 2e-2 tolerance on the delta was covering for.)
        g64 = ref_sd64[name].grad
        gscale = float(max(v.grad.norm() for k2, v in ref_sd64.items() if v.grad is not None))
        f32_equivalent(prm.grad.detach().cpu().numpy(), ref_sd[name].grad.numpy(), g64.numpy(),
                       floor=1e-5 * gscale / (float(g64.norm()) + 1e-30) + 1e-5, what=name)
        # ... and the optimizer applied exactly that gradient
        want = sd0[name] - opt.lr * prm.grad.detach().cpu()
        np.testing.assert_allclose(prm.detach().cpu().numpy(), want.numpy(), rtol=0, atol=2e-7 * float(sd0[name].abs().max()) + 1e-9)
    stats2, _ = trainer.train(2, [dict(batch)])
    assert np.isfinite(stats2["loss"])
    val, _ = trainer.val(2, [dict(batch)])
    assert np.isfinite(val["loss"]) and val["cr_loss"] == 0


def _semi_trainer(lr=1e-3, seed=323):
    from types import SimpleNamespace
    from cet_pick_amd.models.model import create_model
    from cet_pick_amd.synthetic import seeded_state_dict
    from cet_pick_amd.trains.train_factory import train_factory
    heads = {"hm": 1, "proj": 32}
    opt = SimpleNamespace(task="semi", arch="unet_4", pn=False, ge=False, tau=0.1, temp=0.07, thresh=0.5, cr_weight=0.1,
                          num_stacks=1, contrastive=True, device=torch.device("cuda"), num_iters=-1, print_iter=0,
                          hide_data_time=True, exp_id="t", lr=lr, hipgraph=False)
    model = create_model(opt.arch, heads, 32)
    sd0 = seeded_state_dict(model, seed=seed)
    for k in ("hm.weight", "proj.weight"):
        sd0[k] = sd0[k] * 0.3
    model.load_state_dict(sd0)
    trainer = train_factory["semi"](opt, model, torch.optim.SGD(model.parameters(), lr=opt.lr))
    trainer.set_device([0], None, "cuda")
    return opt, heads, model, sd0, trainer


def _semi_batch(b, d, h, w, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(b, d, h, w, generator=g)
    x_aug = x.flip(-1) + 0.05 * torch.randn(b, d, h, w, generator=g)
    gt = torch.full((b, 1, d, h // 2, w // 2), -1.0)
    r = torch.rand(gt.shape, generator=g)
    gt[r < 0.3] = 0.0
    gt[(r >= 0.3) & (r < 0.4)] = 0.6
    gt[r > 0.96] = 1.0
    return x, x_aug, gt


@pytest.mark.parametrize("pairs", [1, 16])
def test_semi_detector_step_at_c5_size_vs_oracle(pairs):
    """BASELINE config C5 AS STATED: one TomoCRSemiTrainer step (unet_4, --contrastive) on `pairs` pairs of 6 x 64 x 64
    crops - 16 per GPU is the configuration, and what tools/bench_detector.py times.  N = pairs * 6 * 32 * 32 voxels per
    view; at 16 pairs the reference's (2N)^2 similarity matrix would be 154 GB.
      pairs = 1  (2N = 12,288): every loss term against the DENSE oracle (the reference's own arithmetic, 0.6 GB), fp32 and -
                 the arbiter - float64.
      pairs = 16 (2N = 196,608): hm_loss, consis_loss against the oracle as they are; the contrastive term against the
                 oracle's row-blocked form (oracle/loss_ref.py::unbiased_con_loss_streamed, pinned to the dense form in
                 tests/test_oracle_losses.py) evaluated in float64 - plain torch matmul / exp blocks on the device the test runs
                 on, no kernel of this repository - plus the row-sum identities of the streaming kernel on sampled rows of
                 THIS batch's features against dense float64 rows."""
    from oracle import loss_ref as OL, unet_ref as OU
    opt, heads, model, sd0, trainer = _semi_trainer()
    x, x_aug, gt = _semi_batch(pairs, 6, 64, 64, seed=40 + pairs)
    batch = {"input": x, "input_aug": x_aug, "hm": gt, "flip_prob": 0.2, "meta": {}}
    with torch.no_grad():
        def oracle(dt, streamed):
            rsd = {k: (v.to(dt) if v.is_floating_point() else v.clone()) for k, v in sd0.items()}
            o1 = OU.tomo_conv_unet_forward(rsd, x.to(dt), 4, heads, training=True)
            o2 = OU.tomo_conv_unet_forward(rsd, x_aug.to(dt), 4, heads, training=True)
            return o1, o2, OL.tomo_cr_semi_loss(o1["hm"], o2["hm"], o1["proj"], o2["proj"], gt.to(dt), 0.2, opt.tau, opt.temp,
                                                opt.thresh, opt.cr_weight, streamed=streamed)
        if pairs == 1:
            _, _, ref = oracle(torch.float32, None)
            o1_64, o2_64, ref64 = oracle(torch.float64, None)
        else:
            _, _, ref = oracle(torch.float32, {"block": 4096, "device": "cuda"})
            o1_64, o2_64, ref64 = oracle(torch.float64, {"block": 2048, "device": "cuda", "dtype": torch.float64})
    stats, _ = trainer.train(1, [dict(batch)])
    assert set(stats) == {"loss", "hm_loss", "cr_loss", "consis_loss", "time"}
    for k, v, v64 in zip(("loss", "hm_loss", "cr_loss", "consis_loss"), ref, ref64):
        # as far from float64 as twice the fp32 oracle, or 1e-4 (the network's outputs differ by fp32 rounding order)
        assert abs(stats[k] - v64.item()) <= 2 * abs(v.item() - v64.item()) + 1e-4 * abs(v64.item()) + 1e-7, (k, stats[k], v.item(), v64.item())
    assert all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in model.parameters())
    if pairs == 16:
        # the streaming kernel's row sums on THIS batch's features (float64 oracle features, rounded to fp32): sampled rows
        from cet_pick_amd.models.loss import _UclRowSumsFn
        ch = o1_64["proj"].shape[1]
        flat = lambda p: p.reshape(pairs, ch, -1).permute(1, 0, 2).reshape(ch, -1).T
        f = torch.cat([flat(o1_64["proj"]), flat(o2_64["proj"].flip(-1))], 0)
        f32 = f.float().cuda().contiguous()
        n2 = f32.shape[0]
        assert n2 == 2 * pairs * 6 * 32 * 32
        cls = torch.randint(0, 4, (n2,), generator=torch.Generator().manual_seed(5)).to(torch.uint8).cuda()
        m, sa, sp, so, ep = _UclRowSumsFn.apply(f32, cls, 1.0 / opt.temp)
        rows = torch.arange(7, n2, 769, device="cuda")
        fd = f32.double()
        S = (fd[rows] @ fd.t()) / opt.temp
        mx = S.max(1, keepdim=True)[0]
        np.testing.assert_allclose(m[rows].cpu().numpy(), mx[:, 0].cpu().numpy(), rtol=1e-5)
        E = torch.exp(S - mx)
        E[torch.arange(rows.numel()), rows] = 1.0
        np.testing.assert_allclose(sa[rows].cpu().numpy(), E.sum(1).cpu().numpy(), rtol=2e-4)
        np.testing.assert_allclose(sp[rows].cpu().numpy(), (E * (cls & 1).double()).sum(1).cpu().numpy(), rtol=2e-4, atol=1e-6)
        np.testing.assert_allclose(so[rows].cpu().numpy(), (E * ((cls >> 1) & 1).double()).sum(1).cpu().numpy(), rtol=2e-4, atol=1e-6)
        np.testing.assert_allclose(ep[rows].cpu().numpy(), E[torch.arange(rows.numel()), (rows + n2 // 2) % n2].cpu().numpy(),
                                   rtol=2e-4, atol=1e-7)
        # ---- the BACKWARD of the streaming kernel at this size (ucl_bwd_kernel<32, 0> row side and
        # <32, 1> column side), on sampled rows against dense float64 rows:
        #   dF[i] = 1/T sum_{j != i} ( c(i; j) E_ij + c(j; i) E_ji ) F[j],  E_ij = exp(S_ij - rowmax_i),
        #   c(i; j) = g_all[i] + g_pos[i] [pos j] + g_other[i] [other j] + g_pair[i] [j = pair(i)]
        # with random upstream gradients for the four differentiable outputs (the row maximum is the kernel's own, checked above)
        gen = torch.Generator(device="cuda").manual_seed(11)
        ups = [torch.rand(n2, device="cuda", generator=gen) * sc for sc in (1.0, 3.0, 2.0, 50.0)]
        fg = f32.clone().requires_grad_()
        _, a2, p2, o2, e2 = _UclRowSumsFn.apply(fg, cls, 1.0 / opt.temp)
        (a2 * ups[0] + p2 * ups[1] + o2 * ups[2] + e2 * ups[3]).sum().backward()
        dF = fg.grad
        rows = torch.arange(5, n2, 2311, device="cuda")                             # 86 rows from both halves
        ar = torch.arange(rows.numel(), device="cuda")
        md = m.double()
        posd, othd = (cls & 1).double(), ((cls >> 1) & 1).double()
        U = [u.double() for u in ups]
        S = (fd[rows] @ fd.t()) / opt.temp                                         # S_ij = S_ji
        pair = (rows + n2 // 2) % n2
        Er = torch.exp(S - md[rows][:, None])                                       # E_ij, row side
        Wr = Er * (U[0][rows][:, None] + U[1][rows][:, None] * posd[None, :] + U[2][rows][:, None] * othd[None, :])
        Wr[ar, pair] += U[3][rows] * Er[ar, pair]
        Ec = torch.exp(S - md[None, :])                                             # E_ji, column side
        Wc = Ec * (U[0][None, :] + U[1][None, :] * posd[rows][:, None] + U[2][None, :] * othd[rows][:, None])
        Wc[ar, pair] += U[3][pair] * Ec[ar, pair]                                   # j with pair(j) = i is j = pair(i)
        W = Wr + Wc
        W[ar, rows] = 0
        want = (W @ fd) / opt.temp
        got = dF[rows].double()
        scale = want.abs().max(1, keepdim=True)[0]
        err = float(((got - want).abs() / scale).max())
        # a row of dF is a sum of 196,608 vectors in all directions: it is ~1/400 of the sum of its terms' magnitudes, so
        # fp32 summation alone leaves ~sqrt(N) eps x 400 ~ 1e-5 .. 1e-3 of the row.  The bound is therefore two-sided: against
        # the magnitude of what is summed (a wrong coefficient, class bit, pair term or a dropped tile is O(1) of it) ...
        mag = ((W.abs() @ fd.abs()) / opt.temp).max(1, keepdim=True)[0]
        assert float(((got - want).abs() / mag).max()) <= 5e-6, float(((got - want).abs() / mag).max())
        # ... and against the same dense rows evaluated in float32 by plain torch: no further from float64 than twice that
        W32 = W.float()
        want32 = ((W32 @ f32) / opt.temp).double()
        err32 = float(((want32 - want).abs() / scale).max())
        assert err <= 2 * err32 + 2e-4, (err, err32)
        # ---- the step's parameter gradients of the layers the loss reaches first (the (3, 1, 1) heads, the dilated 3-D feature
        # head) against the oracle under autograd in float64, its contrastive term in the differentiable blocked form
        # (oracle/loss_ref.py::_StreamedUCL, pinned to autograd through the dense form in tests/test_oracle_losses.py)
        rsd = {k: (v.double().clone().requires_grad_(v.is_floating_point() and k.endswith(".weight") or k.endswith(".bias"))
                   if v.is_floating_point() else v.clone()) for k, v in sd0.items()}
        o1g = OU.tomo_conv_unet_forward(rsd, x.double(), 4, heads, training=True)
        o2g = OU.tomo_conv_unet_forward(rsd, x_aug.double(), 4, heads, training=True)
        tot64 = OL.tomo_cr_semi_loss(o1g["hm"], o2g["hm"], o1g["proj"], o2g["proj"], gt.double(), 0.2, opt.tau, opt.temp, opt.thresh,
                                     opt.cr_weight, streamed={"block": 2048, "device": "cuda", "dtype": torch.float64, "grad": True})[0]
        tot64.backward()
        assert abs(tot64.item() - ref64[0].item()) <= 1e-9 * abs(ref64[0].item())
        checked = 0
        for name, prm in model.named_parameters():
            if not name.startswith(("hm.", "proj.", "feature_head.")):
                continue
            g64 = rsd[name].grad
            assert g64 is not None, name
            g = prm.grad.detach().double().cpu()
            err = float((g - g64).abs().max() / g64.abs().max())
            assert err <= 2e-4, (name, err)
            checked += 1
        assert checked >= 4


def test_symmetric_moco_variant_vs_oracle():
    """MoCoModel(symmetric=True) - EMA first, loss in both directions, one enqueue of both key sets."""
    import numpy as np
    from oracle import train_ref as O
    from cet_pick_amd.models.networks.moco_encoder_3d import get_moco_net_small_3d
    from cet_pick_amd.synthetic import seeded_state_dict
    from cet_pick_amd.trains.tomo_moco_small_trainer import MoCoModel, MoCoTrainer
    from types import SimpleNamespace
    heads = {"proj": 256, "pred": 256}
    eq, ek = get_moco_net_small_3d(18, heads, 0), get_moco_net_small_3d(18, heads, 0)
    sd0 = seeded_state_dict(eq, seed=330)
    for kk in [k for k in sd0 if k.startswith("pred.")]:      # 'pred' re-registers the 'proj' module: one set of weights
        sd0["proj." + kk[5:]] = sd0[kk]
    eq.load_state_dict(sd0)
    ek.load_state_dict(sd0)                  # (MoCoModel copies parameters q -> k, not buffers: the fixture seeded both)
    from cet_pick_amd.synthetic import moco_small_inputs
    im1, im2, queue0 = moco_small_inputs()
    model = MoCoModel(eq, ek, dim=128, K=64, m=0.99, T=0.1, symmetric=True, shuffle=False)
    model.queue.copy_(queue0)
    model = model.cuda().train()
    sd_q = {k: v.clone().requires_grad_(k.endswith(O.PARAM_SUFFIX)) for k, v in sd0.items()}
    sd_k = {k: v.clone() for k, v in sd0.items()}
    ref_loss, ref_k, ref_queue, ref_ptr = O.symmetric_moco_step(sd_q, sd_k, queue0, 0, im1, im2, 0.99, 0.1)
    ref_loss.backward()
    # the same step in float64: the arbiter of the gradient comparison below
    dbl = lambda t: t.double() if t.is_floating_point() else t.clone()
    sd_q64 = {k: dbl(v).clone().requires_grad_(k.endswith(O.PARAM_SUFFIX)) for k, v in sd0.items()}
    sd_k64 = {k: dbl(v) for k, v in sd0.items()}
    loss64 = O.symmetric_moco_step(sd_q64, sd_k64, queue0.double(), 0, im1.double(), im2.double(), 0.99, 0.1)[0]
    loss64.backward()
    from conftest import f32_equivalent
    loss, stats = model(im1.cuda(), im2.cuda())
    loss.backward()
    assert abs(float(loss) - float(ref_loss)) < 2e-4 * max(1.0, abs(float(ref_loss)))
    assert int(model.queue_ptr) == ref_ptr == 16
    np.testing.assert_allclose(model.queue.cpu().numpy(), ref_queue.numpy(), rtol=0, atol=2e-5)
    for name, prm in model.encoder_q.named_parameters():
        if name.startswith("pred."):
            continue
        rg = sd_q[name].grad
        if float(rg.norm()) < 1e-5:          # fc.bias sits in front of a batch-statistics BatchNorm: exactly-zero gradient
            assert float(prm.grad.norm()) < 1e-4, name
            continue
        # stem-level gradients run through two forward passes; fp32 summation order moves them: arbitrated by float64
        f32_equivalent(prm.grad.cpu().numpy(), rg.numpy(), sd_q64[name].grad.numpy(), floor=2e-5, what=name)
    for name, prm in model.encoder_k.named_parameters():
        np.testing.assert_allclose(prm.detach().cpu().numpy(), ref_k[name].numpy(), rtol=0, atol=1e-6)
    # ... and against the reference's own MoCoModel.forward (tests/golden/moco_small.npz, gen_golden.py::gen_moco_small)
    import os
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "moco_small.npz"))
    assert abs(float(loss) - float(G["loss_sym"])) < 2e-4 * float(G["loss_sym"])
    assert int(model.queue_ptr) == int(G["ptr_sym"])
    np.testing.assert_allclose(model.queue.cpu().numpy(), G["queue_sym"], rtol=0, atol=3e-5)
    np.testing.assert_allclose(model.encoder_k.fc.weight.detach().cpu().reshape(-1)[::7].numpy(), G["k_fc_weight_sym"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(model.encoder_k.bn1.running_mean.cpu().numpy(), G["k_bn1_running_mean_sym"], rtol=0, atol=1e-5)
    idx = G["sample_idx"]
    for name, prm in model.encoder_q.named_parameters():
        if name.startswith("pred.") or f"grad_sym_{name}_norm" not in G.files:
            continue
        want = float(G[f"grad_sym_{name}_norm"])
        gf = prm.grad.detach().cpu().reshape(-1).numpy()            # logical (co, ci, kd, kh, kw) order, as the reference
        if want < 1e-5:
            continue
        # the reference ran in fp32 on the CPU: ITS distance from the float64 evaluation is the resolution of the comparison
        # (the stem-level gradients pass through two forward passes: 0.3 % of the largest sample); the GPU may be twice as
        # far from float64 (f32_equivalent above), i.e. three such distances from the reference, + north_star's 1e-3
        g64 = sd_q64[name].grad.reshape(-1).numpy()
        r64 = np.linalg.norm(g64)
        slack = abs(want - r64) / r64
        assert abs(np.linalg.norm(gf.astype(np.float64)) - want) <= (1e-3 + 3 * slack) * want, name
        ws = G[f"grad_sym_{name}_sample"]
        e_ref = float(np.abs(ws - g64[idx % g64.size]).max())
        np.testing.assert_allclose(gf[idx % gf.size], ws, rtol=0, atol=3 * e_ref + 1e-3 * float(np.abs(ws).max()) + 1e-7, err_msg=name)
    # the one-directional variant (symmetric=False, :153-154): loss and the 8 enqueued keys
    ma = MoCoModel(get_moco_net_small_3d(18, heads, 0), get_moco_net_small_3d(18, heads, 0), dim=128, K=64, m=0.99, T=0.1,
                   symmetric=False, shuffle=False)
    ma.encoder_q.load_state_dict(sd0); ma.encoder_k.load_state_dict(sd0)
    ma.queue.copy_(queue0)
    ma = ma.cuda().train()
    la, _ = ma(im1.cuda(), im2.cuda())
    assert abs(float(la) - float(G["loss_asym"])) < 2e-4 * float(G["loss_asym"]) and int(ma.queue_ptr) == int(G["ptr_asym"]) == 8
    np.testing.assert_allclose(ma.queue.cpu().numpy(), G["queue_asym"], rtol=0, atol=3e-5)
    # shuffle on: same loss up to summation order (BatchNorm statistics do not depend on the row order)
    model2 = MoCoModel(get_moco_net_small_3d(18, heads, 0), get_moco_net_small_3d(18, heads, 0), dim=128, K=64, m=0.99,
                       T=0.1, symmetric=True, shuffle=True)
    model2.encoder_q.load_state_dict(sd0); model2.encoder_k.load_state_dict(sd0)
    model2.queue.copy_(queue0)
    model2 = model2.cuda().train()
    loss2, _ = model2(im1.cuda(), im2.cuda())
    assert abs(float(loss2) - float(ref_loss)) < 1e-3 * max(1.0, abs(float(ref_loss)))
    # trainer entry points
    opt = SimpleNamespace(task="moco2d", num_iters=-1, print_iter=0, hide_data_time=True, exp_id="t", lr=1e-3, hipgraph=False)
    tr = MoCoTrainer(opt, model2, torch.optim.SGD(model2.encoder_q.parameters(), lr=1e-3))
    tr.set_device([0], None, "cuda")
    ret, _ = tr.train(1, [{"input": im1, "input_aug": im2}])
    assert set(ret) == {"loss", "moco_loss", "time"} and np.isfinite(ret["loss"])


def _moco_pair(batch_seed, r):
    """MoCo over two seeded moco3d encoders + its state as oracle dictionaries."""
    from cet_pick_amd.models.networks.moco_encoder_3d import get_moco_net_small_3d
    from cet_pick_amd.models.moco import MoCo
    from cet_pick_amd.synthetic import seeded_state_dict
    heads = {"proj": 256, "pred": 256}
    eq, ek = get_moco_net_small_3d(18, heads, 0), get_moco_net_small_3d(18, heads, 0)
    sd0 = seeded_state_dict(eq, seed=317)
    for kk in [k for k in sd0 if k.startswith("pred.")]:
        sd0["proj." + kk[5:]] = sd0[kk]
    eq.load_state_dict(sd0)
    moco = MoCo(eq, ek, dim=128, r=r, m=0.999, T=0.1)
    g = torch.Generator().manual_seed(batch_seed)
    moco.queue.copy_(torch.nn.functional.normalize(torch.randn(128, r, generator=g), dim=0))
    return moco.cuda().train(), g


def _oracle_state(enc):
    return {n: t.detach().cpu().contiguous().clone() for n, t in list(enc.named_parameters()) + list(enc.named_buffers())}


from conftest import gpu_relu_decisions as _gpu_relu_decisions, assert_relu_flips_on_edge


def test_engine_graph_replayed_batch64_step_matches_oracle():
    """The step bench.py times - batch 64, r = 1024, lr 1e-3, replayed from the captured hipGraph, i.e. cached weight images +
    direct kernels + deferred weight-gradient reduce + deferred enqueue + side-stream key branch and weight gradients -
    against oracle/train_ref.MocoRef started from the engine's own state, float64 evaluation of the same oracle as arbiter:
    logits, loss, every parameter gradient, the SGD'd query weights, the EMA'd key weights, queue and pointer
    (models/moco.py:101-146).
    The three calls that capture the graph run at lr 0 (set_lr: the learning rate is a device scalar, the graph follows it), so
    the compared replay is the FIRST weight update from the freshly seeded weights, at the bench's own learning rate.
    Gradients of a network of ReLUs are compared BRANCH BY BRANCH: a batch-64 step has ~1e7 ReLU units, an fp32 forward pass
    is ~1e-6 of an activation's size away from float64, so a few units per step sit closer to zero than that - for them
    "fires or not" is not defined at fp32 resolution, and ONE such unit in layer3 moves every gradient below it by ~1e-2
    (measured: this batch, layer3.0's first ReLU; round 3 covered it with a 5e-3 + 2 x 3e-2 allowance).  So the float64 oracle
    takes the GPU's own decisions (hipops.RELU_TAP: the post-ReLU activations of the same forward pass) for the trunk's
    blocks, feature_3d and the head, and the test checks (1) every unit where that differs from float64's own decision IS on
    the edge (|pre-activation| <= 1e-4 of the layer's rms, and at most a handful per layer), (2) on that branch every gradient
    is within twice the fp32 CPU oracle's distance from float64 + 2e-4 of its norm, and every gradient NORM within 2e-4
    (measured: 1e-5 .. 8e-5 on every parameter, three edge units in the step)."""
    import numpy as np
    from oracle import train_ref as T
    from cet_pick_amd.trains.moco_engine import MocoStepEngine
    moco, g = _moco_pair(5, 1024)
    LR = 1e-3
    eng = MocoStepEngine(moco, lr=0.0, use_graph=True)
    B = 64
    def batch():
        a = torch.randn(B, 1, 32, 32, 32, generator=g)
        return a, a.flip(4) + 0.1 * torch.randn(B, 1, 32, 32, 32, generator=g)
    q_seeded = eng.arena_q.flat.clone()
    for _ in range(3):                                        # eager, eager, capture + first replay: no weight moves
        a, b = batch()
        eng.step(a.cuda(), b.cuda())
    assert eng._graph is not None and eng.node_counts()["kernel"] > 50
    assert torch.equal(eng.arena_q.flat, q_seeded)
    eng.set_lr(LR)
    torch.cuda.synchronize()
    sd_q, sd_k = _oracle_state(moco.encoder_q), _oracle_state(moco.encoder_k)
    queue0, ptr0 = moco.queue.cpu().clone(), int(moco.queue_ptr)
    q_before = eng.arena_q.flat.clone()
    im_q, im_k = batch()
    masks = _gpu_relu_decisions(moco.encoder_q, im_q.cuda())   # same weights, same input, same kernels as the step below
    assert len(masks) == 15
    loss = eng.step(im_q.cuda(), im_k.cuda())                 # ONE graph replay
    torch.cuda.synchronize()

    def run_ref(dt, eps=0.0, relu_masks=None, pre=None):
        cv = lambda t: t.to(dt) if t.is_floating_point() else t.clone()
        ref = T.MocoRef({k: cv(v) for k, v in sd_q.items()}, cv(queue0), m=0.999, T=0.1, lr=LR)
        ref.k = {k: cv(v) for k, v in sd_k.items()}
        ref.ptr = ptr0
        out = ref.step(cv(im_q) * (1.0 + eps), cv(im_k) * (1.0 + eps), pre=pre, relu_masks=relu_masks)
        return ref, out
    # (1) where the GPU's decisions differ from float64's own: only units on the edge
    pre64 = {}
    with torch.no_grad():
        T.encoder_forward({k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd_q.items()}, im_q.double(), True,
                          None, pre64)
    flips = assert_relu_flips_on_edge(masks, pre64)
    # (2) gradients on the GPU's branch
    ref, (lg32, loss32, g32) = run_ref(torch.float32, relu_masks=masks)
    _, (_, _, g32b) = run_ref(torch.float32, eps=2.0 ** -23, relu_masks=masks)   # the same step, inputs one rounding away
    ref64, (lg64, loss64, g64) = run_ref(torch.float64, relu_masks=masks)
    lg = eng.logits.cpu().numpy()
    assert lg.shape == (B, 1025)
    np.testing.assert_allclose(0.1 * lg, 0.1 * lg64.numpy(), rtol=0, atol=1e-3)          # cosines: north_star's 1e-3
    from conftest import f32_equivalent
    f32_equivalent(lg, lg32.numpy(), lg64.numpy(), what="logits")
    assert abs(float(loss) - loss64) <= max(2 * abs(loss32 - loss64) + 2e-5, 2e-4)
    gscale = float(sum(float(v.norm()) ** 2 for v in g64.values()) ** 0.5)
    checked, table = 0, []
    for n, p in moco.encoder_q.named_parameters():
        if n == "fc.bias" or n not in g32:
            continue
        a = p._mi_grad_view.detach().cpu().contiguous().numpy()          # the arena the graph wrote (and SGD consumed)
        n64 = float(g64[n].norm())
        tiny = 5e-5 * gscale / (n64 + 1e-30)                              # (a gradient that is rounding noise of the whole)
        r64 = g64[n].numpy().astype(np.float64)
        e_g = float(np.linalg.norm(a.astype(np.float64) - r64)) / (n64 + 1e-30)
        e_c = max(float(np.linalg.norm(x_[n].numpy().astype(np.float64) - r64)) / (n64 + 1e-30) for x_ in (g32, g32b))
        table.append((n, e_g, e_c, tiny, abs(float(np.linalg.norm(a.astype(np.float64))) - n64) / (n64 + 1e-30)))
        checked += 1
    assert checked >= 28
    if os.environ.get("CETPICK_TEST_VERBOSE"):
        print("units whose decision differs from float64's:", {k: v for k, v in flips.items() if v})
        for n, e_g, e_c, tiny, e_n in table:
            print("%-28s gpu %.2e  cpu32 (worst of 2) %.2e   norm %.2e" % (n, e_g, e_c, e_n))
    for n, e_g, e_c, tiny, e_n in table:
        assert e_g <= 2 * e_c + 2e-4 + tiny, "grad %s: GPU %.3e from float64, CPU fp32 %.3e" % (n, e_g, e_c)
        assert e_n <= 2e-4 + tiny, "norm of grad %s: %.3e" % (n, e_n)
    # SGD applied exactly the arena's gradient; EMA, queue and pointer follow the oracle
    want_q = q_before - LR * eng.arena_q.flat_grad
    assert float((eng.arena_q.flat - want_q).abs().max()) <= 2e-7 * float(q_before.abs().max())
    for n, p in moco.encoder_k.named_parameters():
        np.testing.assert_allclose(p.detach().cpu().contiguous().numpy(), ref.k[n].numpy(), rtol=0, atol=2e-6, err_msg=n)
    np.testing.assert_allclose(moco.queue.cpu().numpy(), ref.queue.numpy(), rtol=0, atol=2e-4)
    assert int(moco.queue_ptr) == ref.ptr == (ptr0 + B) % 1024
    np.testing.assert_allclose(moco.encoder_q.bn1.running_var.cpu().numpy(), ref.q["bn1.running_var"].numpy(), rtol=1e-4, atol=1e-6)
    eng.close()


def test_engine_step_equals_plain_sequence_bitwise():
    """the fast paths that exist only inside MocoStepEngine - cached pre-cut weight images, the deferred
    split-K weight-gradient reduce, the deferred enqueue on a stable queue, the loss buffer - against the plain sequence
    moco(); cross_entropy_label0; backward; sgd_step_ from the same state.  Both run the same kernels in the same
    order of summation: gradients, both arenas, queue and pointer must be EQUAL BIT FOR BIT on every step - also after
    the weights were written through the flat arena between two steps (a stale image would show here)."""
    from cet_pick_amd import hipops as H
    from cet_pick_amd.trains.moco_engine import MocoStepEngine
    mA, g = _moco_pair(9, 64)
    mB, _ = _moco_pair(9, 64)
    eng = MocoStepEngine(mA, lr=1e-3, use_graph=False)
    aq, ak = mB.flatten_parameters()
    assert torch.equal(eng.arena_q.flat, aq.flat) and torch.equal(mA.queue, mB.queue)
    B = 16
    for step in range(4):
        x = torch.randn(B, 1, 32, 32, 32, generator=g).cuda()
        y = (x.flip(4) + 0.1 * torch.randn(B, 1, 32, 32, 32, generator=g).cuda())
        if step == 2:
            # a write from outside the step, through the ARENA (what dist.broadcast / a checkpoint load do): the engine
            # has to notice and re-cut its images
            with torch.no_grad():
                eng.arena_q.flat.mul_(1.001); aq.flat.mul_(1.001)
                eng.arena_k.flat.mul_(0.999); ak.flat.mul_(0.999)
        la = eng.step(x, y)
        aq.zero_grad()
        logits, _ = mB(x, y)
        lb = H.cross_entropy_label0(logits)
        lb.backward()
        H.sgd_step_(aq.flat, aq.flat_grad, 1e-3)
        torch.cuda.synchronize()
        assert torch.equal(eng.logits, logits.detach()), step
        assert float(la) == float(lb), step
        assert torch.equal(eng.arena_q.flat_grad, aq.flat_grad), step
        assert torch.equal(eng.arena_q.flat, aq.flat) and torch.equal(eng.arena_k.flat, ak.flat), step
        assert torch.equal(mA.queue, mB.queue) and int(mA.queue_ptr) == int(mB.queue_ptr) == ((step + 1) * B) % 64, step
    # and a fresh engine started from the plain model's state takes the same next step
    eng2 = MocoStepEngine(mB, lr=1e-3, use_graph=False)
    x = torch.randn(B, 1, 32, 32, 32, generator=g).cuda()
    eng.step(x, x.flip(3)); eng2.step(x, x.flip(3))
    torch.cuda.synchronize()
    assert torch.equal(eng.arena_q.flat, eng2.arena_q.flat) and torch.equal(eng.arena_q.flat_grad, eng2.arena_q.flat_grad)
