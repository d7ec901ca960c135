"""Run the inference hot paths a few times (for rocprofv3 --kernel-trace --stats / --pmc).
usage: prof_infer.py [decode|dog|both] [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cet_pick_amd.synthetic import make_tomo, make_logits
from cet_pick_amd.models import decode as Dm
from cet_pick_amd.utils import image as Im

what = sys.argv[1] if len(sys.argv) > 1 else "both"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
if what in ("decode", "both"):
    logits = torch.as_tensor(make_logits((128, 256, 256), seed=317)).cuda()[None, None]
    for _ in range(reps):
        Dm.sigmoid_tomo_decode(logits, kernel=3, K=900)
    torch.cuda.synchronize()
if what in ("dog", "both"):
    vol, _ = make_tomo((256, 512, 512), seed=317)
    v = torch.as_tensor(vol).cuda()
    for _ in range(reps):
        Im.dog_pick(v, [3, 5])
    torch.cuda.synchronize()
print("done")
