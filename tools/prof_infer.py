"""Run the inference hot paths a few times (for rocprofv3 --kernel-trace --stats / --pmc).
usage: prof_infer.py [decode|dog|both] [reps]     (reps + 2 calls of each chain are launched: see region())"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cet_pick_amd.synthetic import make_tomo, make_logits
from cet_pick_amd.models import decode as Dm
from cet_pick_amd.utils import image as Im

ARGS = [a for a in sys.argv[1:] if not a.startswith("--")]
what = ARGS[0] if len(ARGS) > 0 else "both"
reps = int(ARGS[1]) if len(ARGS) > 1 else 10


def region(fn):
    """two calls with a synchronize behind each, then `reps` back to back: the very first launch of a chain that is queued
    behind other work without a synchronize once cost rounds_all_kernel 767 us instead of 45 (profiles/r04_experiments.txt item 15)"""
    for _ in range(2):
        fn()
        torch.cuda.synchronize()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()


if what in ("decode", "both"):
    logits = torch.as_tensor(make_logits((128, 256, 256), seed=317)).cuda()[None, None]
    region(lambda: Dm.sigmoid_tomo_decode(logits, kernel=3, K=900))
if what in ("dog", "both"):
    vol, _ = make_tomo((256, 512, 512), seed=317)
    v = torch.as_tensor(vol).cuda()
    region(lambda: Im.dog_pick(v, [3, 5]))
print("done")
