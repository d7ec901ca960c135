"""Timing-build helper: runs the picker's filter stage a few times (results of a DOGF_DBG build are garbage: the greedy tail is
not what is measured - MI_DOG_MAXOUT keeps it bounded)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cet_pick_amd.synthetic import make_tomo
from cet_pick_amd.utils import image as Im
vol, _ = make_tomo((256, 512, 512), seed=317)
v = torch.as_tensor(vol).cuda()
for _ in range(3):
    Im.dog_pick(v, [3, 5])
torch.cuda.synchronize()
print("done")
