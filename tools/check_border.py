"""A/B check of the border-class schedules against the plain schedule (MI_CONV_NO_BORDER=1) on random shapes."""
import os, sys, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cet_pick_amd import hipops as H

def run(case, seed):
    n, d, h, w, ci, co, k, s, p = case
    g = torch.Generator(device="cuda").manual_seed(seed)
    x = torch.randn(n, d, h, w, ci, device="cuda", generator=g)
    wt = H.conv_weight_param(co, ci, k)
    wt.data = (torch.randn(wt.shape, device="cuda", generator=g) * 0.1).permute(2, 3, 4, 1, 0).contiguous().permute(4, 3, 0, 1, 2)
    res = None
    outs = []
    for env in ("1", ""):
        if env: os.environ["MI_CONV_NO_BORDER"] = env
        else: os.environ.pop("MI_CONV_NO_BORDER", None)
        y = H.conv_fwd(x, wt, k, s, p, relu=True)
        dy = torch.randn(y.shape, device="cuda", generator=torch.Generator(device="cuda").manual_seed(seed + 1))
        dx = H.conv_dgrad(dy, wt, tuple(x.shape), k, s, p, res=x, mask=x)
        wt.grad = None
        H.conv_wgrad_into(x, dy, wt, k, s, p)
        outs.append((y, dx, wt.grad.clone()))
    errs = [float((a - b).abs().max() / (b.abs().max() + 1e-9)) for a, b in zip(outs[1], outs[0])]
    return errs

cases = [(64, 8, 8, 8, 64, 64, 3, 1, 1), (64, 4, 4, 4, 128, 128, 3, 1, 1), (64, 2, 2, 2, 256, 256, 3, 1, 1),
         (64, 8, 8, 8, 64, 128, 3, 2, 1), (64, 4, 4, 4, 128, 256, 3, 2, 1), (3, 5, 6, 7, 32, 64, 3, 1, 1),
         (7, 3, 4, 2, 64, 32, 3, 1, 1), (2, 1, 9, 9, 64, 64, 3, 1, 1), (5, 4, 4, 4, 16, 32, 3, 1, 1), (64, 2, 2, 2, 256, 256, 1, 1, 0)]
bad = 0
for i, c in enumerate(cases):
    e = run(c, 100 + i)
    ok = all(v < 2e-5 for v in e)
    bad += not ok
    print(c, ["%.1e" % v for v in e], "OK" if ok else "MISMATCH")
print("bad:", bad)
sys.exit(1 if bad else 0)
