"""Matrix-core picker chain (infer_dogm.hip) against the vector chain (MI_NO_DOGM=1) on one volume: NMS'd heat-maps, cutoff, picks."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from cet_pick_amd.synthetic import make_tomo
from cet_pick_amd.utils import image as Im
shape = tuple(int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (64, 256, 256)
vol, _ = make_tomo(shape, seed=321)
v = torch.as_tensor(vol).cuda()
def run():
    s, c, n, cut, heat = Im.dog_pick(v, [3, 5], return_heat=True)
    k = int(n.item())
    return s[:k].cpu().numpy(), c[:k].cpu().numpy(), float(cut.item()), heat.cpu().numpy()
os.environ["MI_DOGM"] = "1"
s0, c0, cut0, h0 = run()
del os.environ["MI_DOGM"]
s1, c1, cut1, h1 = run()
print("picks", len(s0), len(s1), "cutoff", cut0, cut1)
d = np.abs(h0 - h1)
print("heat max abs diff", d.max(), "nonzero", (h0 != 0).sum(), (h1 != 0).sum(), "support differs at", ((h0 != 0) != (h1 != 0)).sum())
if d.max() > 1e-4:
    i = np.unravel_index(np.argmax(d), d.shape); print("worst at", i, h0[i], h1[i])
    nz = np.argwhere((h0 != 0) != (h1 != 0))[:10]; print(nz)
print("coords equal", np.array_equal(c0, c1))
