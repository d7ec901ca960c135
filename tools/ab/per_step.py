import os, sys, time
sys.path.insert(0, "/root/repo")
import torch
from cet_pick_amd import hipops as H
from cet_pick_amd.models.networks.moco_encoder_3d import get_moco_net_small_3d
from cet_pick_amd.models.moco import MoCo
from cet_pick_amd.trains.moco_engine import MocoStepEngine
torch.manual_seed(5)
heads = {"proj": 256, "pred": 256}
moco = MoCo(get_moco_net_small_3d(18, heads, 0), get_moco_net_small_3d(18, heads, 0), dim=128, r=1024, m=0.99, T=0.1).cuda()
moco.train()
eng = MocoStepEngine(moco, lr=1e-3, use_graph=True)
x = torch.randn(64, 1, 32, 32, 32, device="cuda"); y = x.flip(4).contiguous()
for _ in range(5):
    eng.step(x, y)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(61)]
ev[0].record()
for i in range(60):
    eng.step(x, y); ev[i + 1].record()
torch.cuda.synchronize()
ts = [ev[i].elapsed_time(ev[i + 1]) for i in range(60)]
print(" ".join("%.3f" % t for t in ts))
print("first20 %.4f  last20 %.4f" % (sum(ts[:20]) / 20, sum(ts[-20:]) / 20))
