"""bf16x3 conv path (MI_CONV_ARITH=bf16x3, the default) against the native f32 MFMA path (MI_CONV_ARITH=f32) and against float64, on assorted shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from cet_pick_amd import hipops as H


def run(case, seed):
    n, d, h, w, ci, co, k, s, p = case
    g = torch.Generator(device="cuda").manual_seed(seed)
    x = torch.randn(n, d, h, w, ci, device="cuda", generator=g)
    wt = H.conv_weight_param(co, ci, k)
    wt.data = (torch.randn(wt.shape, device="cuda", generator=g) * 0.1).permute(2, 3, 4, 1, 0).contiguous().permute(4, 3, 0, 1, 2)
    # float64 reference (NCDHW)
    x64 = x.permute(0, 4, 1, 2, 3).double().requires_grad_(True)
    w64 = wt.detach().double().requires_grad_(True)
    y64 = F.conv3d(x64, w64, stride=s, padding=p)
    dy = torch.randn(y64.shape, device="cuda", generator=g)
    gx, gw = torch.autograd.grad(y64, (x64, w64), dy.double())
    dy_cl = dy.permute(0, 2, 3, 4, 1).contiguous()
    ref = (y64.permute(0, 2, 3, 4, 1), gx.permute(0, 2, 3, 4, 1), gw)
    errs = []
    for env in ("f32", "bf16x3"):
        os.environ["MI_CONV_ARITH"] = env
        y = H.conv_fwd(x, wt, k, s, p)
        dx = H.conv_dgrad(dy_cl, wt, tuple(x.shape), k, s, p) if ci > 1 else ref[1].float()   # the stem has no dgrad
        wt.grad = None
        H.conv_wgrad_into(x, dy_cl, wt, k, s, p)
        outs = (y, dx, wt.grad.clone())
        errs.append([float((a.double() - b).abs().max() / (b.abs().max() + 1e-30)) for a, b in zip(outs, ref)])
    os.environ.pop("MI_CONV_ARITH", None)
    return errs


cases = [(64, 8, 8, 8, 64, 64, 3, 1, 1), (64, 4, 4, 4, 128, 128, 3, 1, 1), (64, 2, 2, 2, 256, 256, 3, 1, 1),
         (64, 8, 8, 8, 64, 128, 3, 2, 1), (64, 4, 4, 4, 128, 256, 3, 2, 1), (3, 5, 6, 7, 32, 64, 3, 1, 1),
         (7, 3, 4, 2, 64, 32, 3, 1, 1), (2, 1, 9, 9, 64, 64, 3, 1, 1), (5, 4, 4, 4, 16, 32, 3, 1, 1),
         (5, 4, 4, 4, 48, 16, 3, 1, 1), (64, 2, 2, 2, 256, 256, 1, 1, 0), (16, 2, 2, 2, 256, 512, 1, 2, 0),
         (4, 32, 32, 32, 1, 64, 7, 2, 3), (2, 16, 8, 24, 1, 64, 7, 2, 3)]
bad = 0
print("errors relative to float64 (max |diff| / max |ref|): fwd, dgrad, wgrad")
for i, c in enumerate(cases):
    e32, e3 = run(c, 200 + i)
    ok = all(b < max(4 * a, 2e-6) for a, b in zip(e32, e3))
    bad += not ok
    print(c, "f32", ["%.1e" % v for v in e32], "bf16x3", ["%.1e" % v for v in e3], "OK" if ok else "WORSE")
print("bad:", bad)
sys.exit(1 if bad else 0)
