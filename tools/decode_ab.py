"""Decode chain timing on the benchmark volume (eager + per-kernel breakdown is in tools/gprof.sh)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cet_pick_amd.synthetic import make_logits
from cet_pick_amd.models import decode as Dm
logits = torch.as_tensor(make_logits((128, 256, 256), seed=317)).cuda()[None, None]
f = lambda: Dm.sigmoid_tomo_decode(logits, kernel=3, K=900)
for _ in range(5): f()
torch.cuda.synchronize()
best = 1e9
for rep in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / 20)
print("decode %s: %.2f us" % (os.environ.get("TAG", ""), best * 1e3))
