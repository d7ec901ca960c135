"""stem 7^3 stride-2 convolution: forward / weight gradient timings (hipGraph replay) and error against float64."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from cet_pick_amd import hipops as H

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
x = torch.randn(N, 32, 32, 32, 1, device="cuda")
wt = H.conv_weight_param(64, 1, 7); wt.data = wt.data.cuda(); wt.data.normal_()
y = H.conv_fwd(x, wt, 7, 2, 3)
dy = torch.randn_like(y)
def wg():
    wt.grad = None
    H.conv_wgrad_into(x, dy, wt, 7, 2, 3)
print("N=%d fwd %.1f us  wgrad %.1f us" % (N, timeit(lambda: H.conv_fwd(x, wt, 7, 2, 3)), timeit(wg)))
# accuracy of the weight gradient against float64 (small batch)
xs, dys = x[:4].double(), dy[:4].double()
wr = wt.detach().double().clone().requires_grad_(True)
F.conv3d(xs.permute(0, 4, 1, 2, 3), wr, stride=2, padding=3).backward(dys.permute(0, 4, 1, 2, 3))
wt.grad = None
H.conv_wgrad_into(x[:4].contiguous(), dy[:4].contiguous(), wt, 7, 2, 3)
print("wgrad rel err vs float64: %.2e" % float((wt.grad.double() - wr.grad).norm() / wr.grad.norm()))
