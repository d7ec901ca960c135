"""Quick timing of the inference kernels at BASELINE config-3 sizes (not the driver's bench)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cet_pick_amd.synthetic import make_tomo, make_logits
from cet_pick_amd.models import decode as Dm
from cet_pick_amd.utils import image as Im

def timeit(fn, n=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

logits = torch.as_tensor(make_logits((128, 256, 256), seed=317)).cuda()[None, None]
ms = timeit(lambda: Dm.sigmoid_tomo_decode(logits, kernel=3, K=900))
nv = logits.numel()
print("decode 128x256x256: %.3f ms  %.1f Gvox/s  %.1f GB/s(8B/vox)" % (ms, nv / ms / 1e6, nv * 8 / ms / 1e6))
D, H, W = (256, 512, 512) if len(sys.argv) < 2 else tuple(int(v) for v in sys.argv[1:4])
vol, _ = make_tomo((D, H, W), seed=317)
v = torch.as_tensor(vol).cuda()
g = lambda: Im.gaussian_filter(v, 3.0)
ms = timeit(g, n=5, warm=2)
print("gauss sigma3 %dx%dx%d: %.3f ms  %.1f GB/s (24B/vox)" % (D, H, W, ms, v.numel() * 24 / ms / 1e6))
ms = timeit(lambda: Im.gaussian_filter(v, 5.0), n=5, warm=2)
print("gauss sigma5: %.3f ms  %.1f GB/s (24B/vox)" % (ms, v.numel() * 24 / ms / 1e6))
f = lambda: Im.dog_pick(v, [3, 5])
ms = timeit(f, n=5, warm=2)
s, c, n, cut, _ = f()
print("dog_pick: %.3f ms  %.2f Gvox/s  picks=%d cutoff=%.5f" % (ms, v.numel() / ms / 1e6, int(n.item()), float(cut.item())))
