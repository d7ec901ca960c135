"""A/B of the DoG picker chain on the benchmark volume: dog_pick ms for the environment it is started in."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cet_pick_amd.synthetic import make_tomo
from cet_pick_amd.utils import image as Im
vol, _ = make_tomo((256, 512, 512), seed=317)
v = torch.as_tensor(vol).cuda()
f = lambda: Im.dog_pick(v, [3, 5])
for _ in range(3): f()
torch.cuda.synchronize()
best = 1e9
for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / 10)
s, c, n, cut, _ = f()
print("dog_pick %s: %.4f ms picks=%d cutoff=%.6f" % (os.environ.get("TAG", ""), best, int(n.item()), float(cut.item())))
