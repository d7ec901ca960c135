"""The picker queued behind decodes WITHOUT a synchronize (the first launch of its tail in a process): per-call durations of
rounds_all_kernel from a rocprofv3 kernel trace (profiles/r04_experiments.txt item 15)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cet_pick_amd.synthetic import make_tomo, make_logits
from cet_pick_amd.models import decode as Dm
from cet_pick_amd.utils import image as Im
logits = torch.as_tensor(make_logits((128, 256, 256), seed=317)).cuda()[None, None]
vol, _ = make_tomo((256, 512, 512), seed=317)
v = torch.as_tensor(vol).cuda()
for _ in range(10):
    Dm.sigmoid_tomo_decode(logits, kernel=3, K=900)
for _ in range(6):
    Im.dog_pick(v, [3, 5])
torch.cuda.synchronize()
print("done")
