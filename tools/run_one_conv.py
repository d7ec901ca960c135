"""Launch one conv layer/mode repeatedly (for rocprofv3 --pmc runs)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cet_pick_amd import hipops as H
from tools.bench_conv import LAYERS  # noqa
name, mode = sys.argv[1], sys.argv[2]
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
n, d, h, w, ci, co, k, s, p = LAYERS[name]
x = torch.randn(n, d, h, w, ci, device="cuda")
wt = H.conv_weight_param(co, ci, k); wt.data = wt.data.cuda(); wt.data.normal_()
y = H.conv_fwd(x, wt, k, s, p)
dy = torch.randn_like(y)
for _ in range(reps):
    if mode == "fwd": H.conv_fwd(x, wt, k, s, p)
    elif mode == "dgrad": H.conv_dgrad(dy, wt, x.shape, k, s, p)
    else:
        wt.grad = None
        H.conv_wgrad_into(x, dy, wt, k, s, p)
torch.cuda.synchronize()
print("done")
