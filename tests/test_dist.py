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
    """the engine's bucketed gradient exchange on CPU tensors over gloo: four asynchronous all-reduces over disjoint arena
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
    for w in eng._pending:
        w.wait()
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
