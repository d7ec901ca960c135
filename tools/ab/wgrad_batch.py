"""Batched weight gradients of a stage alone (hipops.run_wgrad_jobs + the slab reduce), us per stage:
    python tools/ab/wgrad_batch.py layer2|layer1|layer3 [batch]       (A/B by environment: MI_NO_D3S_WGRAD, MI_D3SW_BATCH_SPLITS, ...)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cet_pick_amd import hipops as H

stage = sys.argv[1] if len(sys.argv) > 1 else "layer2"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
d, c, nb = {"layer1": (8, 64, 4), "layer2": (4, 128, 3), "layer3": (2, 256, 3)}[stage]
g = torch.Generator(device="cuda").manual_seed(1)
xs = [torch.randn(B, d, d, d, c, device="cuda", generator=g) for _ in range(nb)]
dys = [torch.randn(B, d, d, d, c, device="cuda", generator=g) for _ in range(nb)]
ws = []
for _ in range(nb):
    w = H.conv_weight_param(c, c, 3); w.data = w.data.cuda(); ws.append(w)


def stage_once():
    H.DEFERRED_WGRADS = []
    H.SIDE_WGRADS = []
    for x, dy, w in zip(xs, dys, ws):
        w.grad = None
        H.conv_wgrad_into(x, dy, w, 3, 1, 1)
    H.run_wgrad_jobs(H.SIDE_WGRADS)
    H.SIDE_WGRADS = None
    H.flush_wgrad_reduces()
    H.DEFERRED_WGRADS = None


for _ in range(3):
    stage_once()
torch.cuda.synchronize()
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    for _ in range(4):
        stage_once()
for _ in range(3):
    gr.replay()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    gr.replay()
e1.record()
torch.cuda.synchronize()
from cet_pick_amd import _lib as L
print("%s x %d, batch %d: %.1f us per stage (launch + reduce)   [%s]" % (stage, nb, B, e0.elapsed_time(e1) / 40 * 1e3,
                                                                      L.lib().mi_debug_last_conv_kernel().decode()), flush=True)
