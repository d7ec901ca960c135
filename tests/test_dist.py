"""world_size-2 tests of the data-parallel exchange steps.

CPU (gloo): the collectives' plumbing (`concat_all_gather`, rank ordering).
GPU (-m gpu): two ranks sharing the one MI355X over gloo run SyncBN and a full MoCo step; the result
must equal ONE process running the concatenated batch (that is what SyncBN + key all-gather +
gradient averaging are for).  On the 8-GPU node the same code runs over RCCL (backend "nccl").
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _init(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    if REPO not in sys.path:
        sys.path.insert(0, REPO)
    dist.init_process_group("gloo", rank=rank, world_size=world)


def _w_gather(rank, world, port, out):
    _init(rank, world, port)
    from cet_pick_amd.models.moco import concat_all_gather
    t = torch.full((3, 4), float(rank + 1))
    t[0, 0] = 10 * rank
    g = concat_all_gather(t)
    s = t.clone()
    dist.all_reduce(s)
    torch.save({"g": g, "s": s}, os.path.join(out, "r%d.pt" % rank))
    dist.destroy_process_group()


def test_concat_all_gather_gloo_cpu(tmp_path):
    port = _free_port()
    mp.spawn(_w_gather, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(str(tmp_path / "r0.pt"))
    r1 = torch.load(str(tmp_path / "r1.pt"))
    assert torch.equal(r0["g"], r1["g"]) and r0["g"].shape == (6, 4)
    assert float(r0["g"][0, 0]) == 0 and float(r0["g"][3, 0]) == 10        # rank-major order
    assert float(r0["g"][1, 1]) == 1 and float(r0["g"][4, 1]) == 2
    assert torch.equal(r0["s"], r1["s"]) and float(r0["s"][1, 1]) == 3


def _w_buckets(rank, world, port, out):
    """the engine's bucketed gradient exchange on CPU tensors over gloo: four all-reduces over disjoint arena
    ranges, issued deepest layers first, must equal one all-reduce of the whole arena"""
    _init(rank, world, port)
    from cet_pick_amd.models.networks.moco_encoder_3d import TomoResClassifier3D, BasicBlock
    from cet_pick_amd.models.moco import MoCo
    from cet_pick_amd.trains.moco_engine import MocoStepEngine
    torch.manual_seed(1)
    heads = {"proj": 256, "pred": 256}
    moco = MoCo(TomoResClassifier3D(BasicBlock, [2, 2, 2, 2], heads, 0), TomoResClassifier3D(BasicBlock, [2, 2, 2, 2], heads, 0),
                dim=128, r=64, m=0.99, T=0.1)
    eng = MocoStepEngine(moco, lr=0.1)                       # CPU tensors: only the exchange plumbing is exercised
    ranges = sorted(eng._bucket.values())
    assert ranges[0][0] == 0 and ranges[-1][1] == eng.arena_q.numel
    assert all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))                   # a partition of the arena
    names = dict(zip([n for n, _ in moco.encoder_q.named_parameters()], eng.arena_q.offsets))
    assert eng._bucket["stem"][1] == names["layer1.0.conv1.weight"] and eng._bucket["layer3"][0] == names["layer3.0.conv1.weight"]
    assert moco.encoder_q.grad_marker is not None
    g = torch.Generator().manual_seed(10 + rank)
    eng.arena_q.flat_grad.copy_(torch.randn(eng.arena_q.numel, generator=g))
    whole = eng.arena_q.flat_grad.clone()
    dist.all_reduce(whole)
    for tag in ("layer3", "layer2", "layer1"):               # what the autograd hooks do during backward
        eng._on_marker(tag)
    eng._reduce_bucket("stem")
    assert eng.buckets_sent == ["layer3", "layer2", "layer1", "stem"]
    torch.save({"ok": bool(torch.equal(eng.arena_q.flat_grad, whole))}, os.path.join(out, "b%d.pt" % rank))
    dist.destroy_process_group()


def test_bucketed_gradient_exchange_gloo_cpu(tmp_path):
    port = _free_port()
    mp.spawn(_w_buckets, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert torch.load(str(tmp_path / "b0.pt"))["ok"] and torch.load(str(tmp_path / "b1.pt"))["ok"]


# ------------------------------------------------------------------------------------------- GPU
def _make_moco(seed=317):
    from cet_pick_amd.models.networks.moco_encoder_3d import TomoResClassifier3D, BasicBlock
    from cet_pick_amd.models.moco import MoCo
    from cet_pick_amd.synthetic import seeded_state_dict
    encs = []
    for _ in range(2):
        e = TomoResClassifier3D(BasicBlock, [2, 2, 2, 2], {"proj": 256, "pred": 256}, 0)
        e.load_state_dict(seeded_state_dict(e, seed=seed))
        encs.append(e)
    torch.manual_seed(5)
    m = MoCo(encs[0], encs[1], dim=128, r=64, m=0.99, T=0.1).cuda()
    return m


def _batches():
    g = torch.Generator().manual_seed(77)
    xq = torch.randn(8, 1, 32, 32, 32, generator=g)
    xk = xq.flip(4) + 0.1 * torch.randn(8, 1, 32, 32, 32, generator=g)
    return xq, xk


def _w_step(rank, world, port, out):
    _init(rank, world, port)
    torch.cuda.set_device(0)
    from cet_pick_amd import hipops as H
    from cet_pick_amd.trains.moco_engine import MocoStepEngine
    moco = _make_moco()
    H.convert_sync_batchnorm(moco)
    moco.train()
    moco.pair_sync_bn = True                         # (round 6, opt-in: one SyncBN collective per layer for both encoders)
    eng = MocoStepEngine(moco, lr=0.05)
    xq, xk = _batches()
    sl = slice(4 * rank, 4 * rank + 4)
    # SyncBN alone: forward + backward of one layer on this rank's half
    bn = H.HipBatchNorm(64).cuda()
    bn.sync = True
    g = torch.Generator().manual_seed(3)
    xb = torch.randn(8, 4, 4, 4, 64, generator=g)
    dyb = torch.randn(8, 4, 4, 4, 64, generator=g)
    xin = xb[sl].cuda().requires_grad_(True)
    yb = bn(xin, relu=True)
    yb.backward(dyb[sl].cuda())
    loss = eng.step(xq[sl].cuda(), xk[sl].cuda())
    torch.cuda.synchronize()
    # the gradient exchange went out in four buckets, deepest layers first (overlapped with the backward pass)
    assert eng.buckets_sent == ["layer3", "layer2", "layer1", "stem"], eng.buckets_sent
    # round 6: the two forward passes ran layer-locked - one SyncBN collective per layer for encoder_q and encoder_k together
    assert moco.sync_collectives == 5, moco.sync_collectives
    # ... and the un-paired form (one collective per layer and branch) gives the same step: a second engine on a fresh copy
    moco_u = _make_moco()
    H.convert_sync_batchnorm(moco_u)
    moco_u.train()
    moco_u.pair_sync_bn = False
    eng_u = MocoStepEngine(moco_u, lr=0.05)
    loss_u = eng_u.step(xq[sl].cuda(), xk[sl].cuda())
    torch.cuda.synchronize()
    assert moco_u.sync_collectives == 0
    assert torch.equal(loss_u, loss) and torch.equal(eng_u.arena_q.flat, eng.arena_q.flat) and torch.equal(eng_u.arena_k.flat, eng.arena_k.flat)
    assert torch.equal(moco_u.queue, moco.queue)
    assert sorted(eng._bucket.values())[0][0] == 0 and sorted(eng._bucket.values())[-1][1] == eng.arena_q.numel
    torch.save({"loss": float(loss), "queue": moco.queue.cpu(), "ptr": int(moco.queue_ptr),
                "q_flat": eng.arena_q.flat.cpu(), "k_flat": eng.arena_k.flat.cpu(),
                "bn_y": yb.detach().cpu(), "bn_dx": xin.grad.cpu(), "bn_dg": bn.weight.grad.cpu(),
                "bn_rm": bn.running_mean.cpu(), "bn1_rv": moco.encoder_q.bn1.running_var.cpu()},
               os.path.join(out, "r%d.pt" % rank))
    dist.destroy_process_group()


@pytest.mark.gpu
def test_two_ranks_equal_one_process_on_the_full_batch(tmp_path):
    from cet_pick_amd import hipops as H
    from cet_pick_amd.trains.moco_engine import MocoStepEngine
    port = _free_port()
    mp.spawn(_w_step, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(str(tmp_path / "r0.pt"))
    r1 = torch.load(str(tmp_path / "r1.pt"))
    # single process, whole batch
    moco = _make_moco()
    moco.train()
    eng = MocoStepEngine(moco, lr=0.05)
    xq, xk = _batches()
    bn = H.HipBatchNorm(64).cuda()
    g = torch.Generator().manual_seed(3)
    xb = torch.randn(8, 4, 4, 4, 64, generator=g)
    dyb = torch.randn(8, 4, 4, 4, 64, generator=g)
    xin = xb.cuda().requires_grad_(True)
    yb = bn(xin, relu=True)
    yb.backward(dyb.cuda())
    loss = eng.step(xq.cuda(), xk.cuda())
    torch.cuda.synchronize()
    # SyncBN == BN over the concatenated batch
    np.testing.assert_allclose(torch.cat([r0["bn_y"], r1["bn_y"]]).numpy(), yb.detach().cpu().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(torch.cat([r0["bn_dx"], r1["bn_dx"]]).numpy(), xin.grad.cpu().numpy(), rtol=1e-4, atol=1e-6)
    # affine gradients are per-rank (as in torch.nn.SyncBatchNorm); their sum is the full-batch gradient
    np.testing.assert_allclose((r0["bn_dg"] + r1["bn_dg"]).numpy(), bn.weight.grad.cpu().numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(r0["bn_rm"].numpy(), bn.running_mean.cpu().numpy(), rtol=1e-5, atol=1e-7)
    # replicas stay identical
    assert torch.equal(r0["queue"], r1["queue"]) and r0["ptr"] == r1["ptr"] == 8
    assert torch.equal(r0["q_flat"], r1["q_flat"]) and torch.equal(r0["k_flat"], r1["k_flat"])
    # ... and equal to the single-process step on the whole batch
    np.testing.assert_allclose(r0["queue"].numpy(), moco.queue.cpu().numpy(), rtol=0, atol=2e-5)
    assert abs(0.5 * (r0["loss"] + r1["loss"]) - float(loss)) < 1e-4
    a, b = r0["q_flat"], eng.arena_q.flat.cpu()
    assert float((a - b).norm()) <= 1e-4 * float(b.norm())
    np.testing.assert_allclose(r0["k_flat"].numpy(), eng.arena_k.flat.cpu().numpy(), rtol=0, atol=1e-5)
    np.testing.assert_allclose(r0["bn1_rv"].numpy(), moco.encoder_q.bn1.running_var.cpu().numpy(), rtol=1e-4, atol=1e-6)


# ------------------------------------------------------------------ f4: DDP shuffle-BN, config 5: GradExchange (CPU, gloo)
def _w_shuffle(rank, world, port, out):
    _init(rank, world, port)
    from cet_pick_amd.models.moco import batch_shuffle_ddp, batch_unshuffle_ddp
    torch.manual_seed(100 + rank)                    # ranks draw different permutations: rank 0's must win
    x = (torch.arange(6, dtype=torch.float32) + 10 * rank).view(6, 1).repeat(1, 3)
    xs, idx_un = batch_shuffle_ddp(x)
    back = batch_unshuffle_ddp(xs * 2.0, idx_un)     # "encode" = x2
    torch.save({"xs": xs, "idx": idx_un, "back": back}, os.path.join(out, "s%d.pt" % rank))
    dist.destroy_process_group()


def test_batch_shuffle_ddp_roundtrip_gloo_cpu(tmp_path):
    """models/moco.py:55-99: the shares of the two ranks are a partition of the global batch under ONE permutation, and
    unshuffle hands every rank its own rows back in order."""
    port = _free_port()
    mp.spawn(_w_shuffle, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    s0, s1 = torch.load(str(tmp_path / "s0.pt")), torch.load(str(tmp_path / "s1.pt"))
    assert torch.equal(s0["idx"], s1["idx"])
    allrows = sorted(torch.cat([s0["xs"][:, 0], s1["xs"][:, 0]]).tolist())
    assert allrows == [0, 1, 2, 3, 4, 5, 10, 11, 12, 13, 14, 15]
    assert not torch.equal(s0["xs"][:, 0], torch.arange(6, dtype=torch.float32))           # actually shuffled
    assert torch.equal(s0["back"][:, 0], 2 * torch.arange(6, dtype=torch.float32))
    assert torch.equal(s1["back"][:, 0], 2 * (torch.arange(6, dtype=torch.float32) + 10))


def _w_exchange(rank, world, port, out):
    _init(rank, world, port)
    from cet_pick_amd import hipops as H
    torch.manual_seed(3)
    net = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.ReLU(), torch.nn.Linear(7, 2), torch.nn.Linear(2, 2))
    ex = H.GradExchange(net, n_buckets=3)
    with torch.no_grad():
        for p in net.parameters():
            p.add_(float(rank))                      # replicas differ until broadcast
    ex.broadcast_parameters(0)
    opt = torch.optim.SGD(net.parameters(), lr=0.1)
    g = torch.Generator().manual_seed(20 + rank)
    x = torch.randn(4, 5, generator=g)
    opt.zero_grad()
    net[:3](x).pow(2).mean().backward()              # the last Linear gets no gradient: zeros, not stale memory
    ex.sync()
    grads = [p.grad.clone() for p in net.parameters()]
    opt.step()
    torch.save({"w": ex.arena.flat.clone(), "g": grads, "x": x}, os.path.join(out, "e%d.pt" % rank))
    dist.destroy_process_group()


def test_grad_exchange_averages_flat_arena_gloo_cpu(tmp_path):
    """hipops.GradExchange (main.py:34-56's DistributedDataParallel replaced by a flat arena + bucketed all-reduce)."""
    port = _free_port()
    mp.spawn(_w_exchange, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    e0, e1 = torch.load(str(tmp_path / "e0.pt")), torch.load(str(tmp_path / "e1.pt"))
    assert torch.equal(e0["w"], e1["w"])                                  # identical replicas after the step
    for a, b in zip(e0["g"], e1["g"]):
        assert torch.equal(a, b)
    torch.manual_seed(3)
    net = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.ReLU(), torch.nn.Linear(7, 2), torch.nn.Linear(2, 2))
    tot = [torch.zeros_like(p) for p in net.parameters()]
    for e in (e0, e1):
        net.zero_grad()
        net[:3](e["x"]).pow(2).mean().backward()
        for t, p in zip(tot, net.parameters()):
            if p.grad is not None:
                t += p.grad / 2
    for t, gavg in zip(tot, e0["g"]):
        np.testing.assert_allclose(gavg.numpy(), t.numpy(), rtol=1e-6, atol=1e-7)


# ------------------------------------------------------------------------------------------- GPU: trainers under DP
def _simsiam_setup():
    from types import SimpleNamespace
    from cet_pick_amd.models.model import create_model
    from cet_pick_amd.synthetic import seeded_state_dict
    net = create_model("simsiam2d_18", {"proj": 128, "pred": 128}, 128)
    net.load_state_dict(seeded_state_dict(net, seed=318))
    opt = SimpleNamespace(task="simsiam3d", num_iters=-1, print_iter=0, hide_data_time=True, exp_id="t", lr=0.05, hipgraph=False)
    g = torch.Generator().manual_seed(9)
    x = torch.randn(8, 1, 36, 36, generator=g)
    return net, opt, x, x.flip(-1) + 0.1 * torch.randn(8, 1, 36, 36, generator=g)


def _w_simsiam(rank, world, port, out):
    _init(rank, world, port)
    torch.cuda.set_device(0)
    from cet_pick_amd import hipops as H
    from cet_pick_amd.trains.train_factory import train_factory
    net, opt, x, xa = _simsiam_setup()
    H.convert_sync_batchnorm(net)
    tr = train_factory["simsiam3d"](opt, net, torch.optim.SGD(net.parameters(), lr=opt.lr))
    tr.set_distributed_device(0)
    assert tr.engine is not None and tr.engine.dist_on and tr.engine.world == 2      # (round 6: the SimSiam step engine does the exchange)
    sl = slice(4 * rank, 4 * rank + 4)
    ret, _ = tr.train(1, [{"input": x[sl], "input_aug": xa[sl]}])
    torch.cuda.synchronize()
    torch.save({"w": tr.engine.arena.flat.cpu(), "loss": ret["loss"]}, os.path.join(out, "m%d.pt" % rank))
    dist.destroy_process_group()


@pytest.mark.gpu
def test_simsiam_trainer_two_ranks_equal_full_batch(tmp_path):
    """simsiam_main.py under torch.distributed: SyncBN + averaged gradients of two half batches == one process on the
    whole batch (the SimSiam loss is a batch mean)."""
    from cet_pick_amd import hipops as H
    from cet_pick_amd.trains.train_factory import train_factory
    port = _free_port()
    mp.spawn(_w_simsiam, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    m0, m1 = torch.load(str(tmp_path / "m0.pt")), torch.load(str(tmp_path / "m1.pt"))
    assert torch.equal(m0["w"], m1["w"])
    net, opt, x, xa = _simsiam_setup()
    w_init = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    tr = train_factory["simsiam3d"](opt, net, torch.optim.SGD(net.parameters(), lr=opt.lr))
    tr.set_device([0], None, "cuda")
    ret, _ = tr.train(1, [{"input": x, "input_aug": xa}])
    arena = H.ParamArena(net)                               # same flat order as the ranks' arenas
    want = arena.flat.cpu()
    assert abs(0.5 * (m0["loss"] + m1["loss"]) - ret["loss"]) < 1e-4
    step = float((want - m0["w"]).norm()) / (float((want - 0).norm()) + 1e-12)
    assert step < 1e-5, step
    assert want.numel() == m0["w"].numel() and w_init.numel() > 0


def _det_setup(seed_batch):
    from types import SimpleNamespace
    from cet_pick_amd.models.model import create_model
    from cet_pick_amd.synthetic import seeded_state_dict
    heads = {"hm": 1, "proj": 32}
    opt = SimpleNamespace(task="semi", arch="unet_4", pn=False, ge=False, tau=0.1, temp=0.07, thresh=0.5, cr_weight=0.1,
                          num_stacks=1, contrastive=True, device=torch.device("cuda"), num_iters=-1, print_iter=0,
                          hide_data_time=True, exp_id="t", lr=1e-3, hipgraph=False)
    model = create_model(opt.arch, heads, 32)
    sd0 = seeded_state_dict(model, seed=323)
    for k in ("hm.weight", "proj.weight"):
        sd0[k] = sd0[k] * 0.3
    model.load_state_dict(sd0)
    g = torch.Generator().manual_seed(seed_batch)
    b, d, h, w = 2, 4, 48, 48
    x = torch.randn(b, d, h, w, generator=g)
    gt = torch.full((b, 1, d, h // 2, w // 2), -1.0)
    r = torch.rand(gt.shape, generator=g)
    gt[r < 0.3] = 0.0
    gt[r > 0.96] = 1.0
    batch = {"input": x, "input_aug": x.flip(-1) + 0.05 * torch.randn(b, d, h, w, generator=g), "hm": gt, "flip_prob": 0.2, "meta": {}}
    return model, opt, batch


def _w_det(rank, world, port, out, same):
    _init(rank, world, port)
    torch.cuda.set_device(0)
    from cet_pick_amd import hipops as H
    from cet_pick_amd.trains.train_factory import train_factory
    model, opt, batch = _det_setup(2 if same else 2 + rank)
    H.convert_sync_batchnorm(model)
    tr = train_factory["semi"](opt, model, torch.optim.SGD(model.parameters(), lr=opt.lr))
    tr.set_distributed_device(0)
    ret, _ = tr.train(1, [dict(batch)])
    torch.cuda.synchronize()
    torch.save({"w": tr.exchange.arena.flat.cpu(), "loss": ret["loss"], "calls": tr.exchange.calls},
               os.path.join(out, "d%d_%d.pt" % (int(same), rank)))
    dist.destroy_process_group()


@pytest.mark.gpu
def test_detector_trainer_data_parallel(tmp_path):
    """main.py:34-56 (BASELINE config 5) through TomoCRSemiTrainer: (1) two ranks fed the SAME crop pairs - SyncBN over
    two identical halves has the statistics of one, the averaged gradient is the gradient - land on the single-process
    step; (2) two ranks with different crop pairs exchange and stay identical replicas."""
    from cet_pick_amd import hipops as H
    from cet_pick_amd.trains.train_factory import train_factory
    for same in (True, False):
        mp.spawn(_w_det, args=(2, _free_port(), str(tmp_path), same), nprocs=2, join=True)
    s0, s1 = torch.load(str(tmp_path / "d1_0.pt")), torch.load(str(tmp_path / "d1_1.pt"))
    d0, d1 = torch.load(str(tmp_path / "d0_0.pt")), torch.load(str(tmp_path / "d0_1.pt"))
    assert s0["calls"] == 1 and torch.equal(s0["w"], s1["w"]) and torch.equal(d0["w"], d1["w"])
    assert d0["loss"] != d1["loss"]                                       # different data per rank
    model, opt, batch = _det_setup(2)
    tr = train_factory["semi"](opt, model, torch.optim.SGD(model.parameters(), lr=opt.lr))
    tr.set_device([0], None, "cuda")
    w0 = H.ParamArena(model)                                              # flat order of the ranks' arenas
    start = w0.flat.clone()
    ret, _ = tr.train(1, [dict(batch)])
    want = w0.flat.cpu()
    upd = float((want - start.cpu()).norm())
    assert upd > 0
    assert float((s0["w"] - want).norm()) <= 2e-3 * upd, (float((s0["w"] - want).norm()), upd)
    assert abs(s0["loss"] - ret["loss"]) <= 1e-4 * abs(ret["loss"])
    assert float((d0["w"] - want).norm()) > 1e-2 * upd                     # other data: another step
