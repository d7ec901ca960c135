"""The stem's BatchNorm + ReLU + MaxPool backward alone (batch 64, 16^3 x 64 channels): the gather form (round 5, MI_POOL_BWD_GATHER=1) against the dense
default form, us per call over 30 calls.   python tools/ab/stem_bwd.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cet_pick_amd import hipops as H

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
g = torch.Generator(device="cuda").manual_seed(3)
x = torch.randn(B, 16, 16, 16, 64, device="cuda", generator=g).requires_grad_(True)
bn = H.HipBatchNorm(64).cuda().train()
y = H.bn_relu_maxpool3d(x, bn, 3, 2, 1)
dy = torch.randn(y.shape, device="cuda", generator=g)
for form in ("gather", "dense", "gather", "dense"):
    if form == "gather":
        os.environ["MI_POOL_BWD_GATHER"] = "1"
    else:
        os.environ.pop("MI_POOL_BWD_GATHER", None)
    for _ in range(5):
        torch.autograd.grad(y, x, dy, retain_graph=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        torch.autograd.grad(y, x, dy, retain_graph=True)
    e1.record()
    torch.cuda.synchronize()
    print("%-7s %.1f us per backward" % (form, e0.elapsed_time(e1) / 30 * 1e3), flush=True)
