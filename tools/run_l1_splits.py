"""l1 forward with forced split counts (kernel-only times come from rocprofv3 --kernel-trace)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cet_pick_amd import hipops as H
n, d, h, w, ci, co, k, s, p = (64, 8, 8, 8, 64, 64, 3, 1, 1)
x = torch.randn(n, d, h, w, ci, device="cuda")
wt = H.conv_weight_param(co, ci, k); wt.data = wt.data.cuda(); wt.data.normal_()
for sp in (1, 2, 3):
    os.environ["MI_CONV_SPLITS"] = str(sp)
    for _ in range(20):
        y = H.conv_fwd(x, wt, k, s, p)
    torch.cuda.synchronize()
