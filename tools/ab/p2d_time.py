"""Time the three compile-time p2d shapes (forward, data gradient, weight gradient) at batch 256: us and TFLOP/s per launch (hipGraph of 8)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cet_pick_amd import hipops as H

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(8): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n / 8 * 1e3

shapes = [(256, 36, 36, 64), (256, 18, 18, 128), (256, 9, 9, 256)] + ([(256, 32, 32, 64), (256, 16, 16, 128), (256, 8, 8, 256)] if "--generic" in sys.argv else [])
for n, h, w, c in shapes:
    x = torch.randn(n, h, w, c, device="cuda"); dy = torch.randn(n, h, w, c, device="cuda")
    p = H.conv2d_weight_param(c, c, 3); p.data = p.data.cuda(); p.data.normal_()
    H.conv_fwd(x, p, 3, 1, 1); H.conv_dgrad(dy, p, tuple(x.shape), 3, 1, 1)
    fl = 2.0 * n * h * w * c * c * 9
    tf = timeit(lambda: H.conv_fwd(x, p, 3, 1, 1)); td = timeit(lambda: H.conv_dgrad(dy, p, tuple(x.shape), 3, 1, 1))
    def wg():
        p.grad = None
        H.conv_wgrad_into(x, dy, p, 3, 1, 1)
    tw = timeit(wg)
    print("%-20s fwd %6.1f us %5.1f TF | dgrad %6.1f us %5.1f TF | wgrad %6.1f us %5.1f TF" % ((n, h, w, c), tf, fl / tf / 1e6, td, fl / td / 1e6, tw, fl / tw / 1e6), flush=True)
