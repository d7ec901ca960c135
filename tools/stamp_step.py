"""Phases of the CAPTURED MoCo step without a profiler attached: one-thread stamp launches (mi_debug_stamp: the 100 MHz wall
clock) are recorded into the graph at the stage boundaries of both forward branches and of the backward pass; the replayed
graph is timed with and without them, and the stamps of the last replays are printed as a timeline.

    python tools/stamp_step.py [batch] [replays]
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cet_pick_amd import hipops as H
from cet_pick_amd.models.networks.moco_encoder_3d import get_moco_net_small_3d
from cet_pick_amd.models.moco import MoCo
from cet_pick_amd.trains.moco_engine import MocoStepEngine

ARGS = [a for a in sys.argv[1:] if not a.startswith("--")]
B = int(ARGS[0]) if len(ARGS) > 0 else 64
R = int(ARGS[1]) if len(ARGS) > 1 else 50
DIST = "--dist" in sys.argv          # the N>1 code path on a 1-rank RCCL group (every collective issued and captured)
if DIST:
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29657")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    H.FORCE_COLLECTIVES = True


def build(stamps):
    torch.manual_seed(5)
    heads = {"proj": 256, "pred": 256}
    moco = MoCo(get_moco_net_small_3d(18, heads, 0), get_moco_net_small_3d(18, heads, 0), dim=128, r=1024, m=0.99, T=0.1).cuda()
    if DIST:
        H.convert_sync_batchnorm(moco)
    moco.train()
    eng = MocoStepEngine(moco, lr=1e-3, use_graph=True)
    if stamps:
        H.STAMPS = (torch.zeros(512, dtype=torch.int64, device="cuda"), [])
        def marker(tag, eng=eng):
            H.stamp("bwd:" + tag)
            eng._on_marker(tag)
        moco.encoder_q.grad_marker = marker
    g = torch.Generator(device="cuda").manual_seed(11)
    x = torch.randn(B, 1, 32, 32, 32, device="cuda", generator=g)
    y = x.flip(4).contiguous()
    for _ in range(6):                       # eager warm-up steps, then the capture
        eng.step(x, y)
    names = None
    if stamps:
        names = list(H.STAMPS[1])            # the names the capture recorded, in slot order
        per = len(names)
    torch.cuda.synchronize()
    return eng, x, y, names


def timed(eng, x, y, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        eng.step(x, y)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


eng, x, y, _ = build(False)
timed(eng, x, y, 20)
print("plain    %.4f ms/step" % timed(eng, x, y, R), flush=True)
print("graph nodes of the captured step: %s; SyncBN collectives of the forward passes: %s (paired: one per layer for encoder_q + encoder_k)"
      % (eng.node_counts(), getattr(eng.moco, "sync_collectives", None)), flush=True)
eng.close()
del eng
H.STAMPS = None

eng, x, y, names = build(True)
buf = H.STAMPS[0]
# the eager warm-up steps and the capture each appended names: keep the LAST occurrence set (the captured one writes the
# highest slots); slot index = position in the list
n_per = None
for i in range(1, len(names)):
    if names[i] == names[0]:
        n_per = i
        break
n_per = n_per or len(names)
base = len(names) - n_per
timed(eng, x, y, 20)
print("stamped  %.4f ms/step  (%d stamps per step)" % (timed(eng, x, y, R), n_per), flush=True)
acc = None
K = 10
STEADY = "--steady" in sys.argv      # stamps of the LAST of six back-to-back replays (the host runs ahead of the device, as in training)
for _ in range(K):
    for _ in range(6 if STEADY else 1):
        eng.step(x, y)
    torch.cuda.synchronize()
    t = buf[base:base + n_per].cpu().double().numpy()
    t = (t - t[0]) / 100.0                # 100 MHz -> us
    acc = t if acc is None else acc + t
acc /= K
order = sorted(range(n_per), key=lambda i: acc[i])
print("   t_us   stamp")
for i in order:
    print("%8.1f  %s" % (acc[i], names[base + i]))
eng.close()
if DIST:
    dist.barrier()
    dist.destroy_process_group()
