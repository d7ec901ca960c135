"""Stride-2 block-front data gradient (csrc/conv_s2.hip) against the two generic launches: us per call, hipGraph replay."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cet_pick_amd import hipops as H

def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(4): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n / 4 * 1e3

for (gi, ci, co) in ((8, 64, 128), (4, 128, 256)):
    n, go = 64, gi // 2
    w = H.conv_weight_param(co, ci, 3); w.data = w.data.cuda(); w.data.normal_()
    wd = H.conv_weight_param(co, ci, 1); wd.data = wd.data.cuda(); wd.data.normal_()
    dh = torch.randn(n, go, go, go, co, device="cuda"); d2 = torch.randn_like(dh)
    mask = torch.randn(n, gi, gi, gi, ci, device="cuda")
    shape = (n, gi, gi, gi, ci)
    fused = lambda: H.conv_dgrad_s2_block(dh, d2, w, wd, shape, None, mask)
    def generic():
        dres = H.conv_dgrad(d2, wd, shape, 1, 2, 0)
        return H.conv_dgrad(dh, w, shape, 3, 2, 1, dres, mask)
    print("grid %d  %d -> %d:  one launch %.1f us   generic %.1f us" % (gi, ci, co, timeit(fused), timeit(generic)))
    # forward of the same front: relu(conv) + shortcut
    x = torch.randn(n, gi, gi, gi, ci, device="cuda")
    def generic_f():
        return H.conv_fwd(x, w, 3, 2, 1, None, True), H.conv_fwd(x, wd, 1, 2, 0)
    for narrow in ("0", "1"):
        os.environ["MI_S2FWD_NARROW"] = narrow
        print("   forward (narrow=%s): one launch %.1f us   generic %.1f us" % (narrow, timeit(lambda: H.conv_fwd_s2_block(x, w, wd)), timeit(generic_f)))
