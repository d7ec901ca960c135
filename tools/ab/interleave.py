"""dog_pick calls interleaved with other work (a decode, a 1 GB fill that evicts L2 / MALL): per-call kernel durations from a
rocprofv3 kernel trace tell whether the picker's tail keeps its steady-state time when it is not called back to back."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cet_pick_amd.synthetic import make_tomo, make_logits
from cet_pick_amd.models import decode as Dm
from cet_pick_amd.utils import image as Im
logits = torch.as_tensor(make_logits((128, 256, 256), seed=317)).cuda()[None, None]
vol, _ = make_tomo((256, 512, 512), seed=317)
v = torch.as_tensor(vol).cuda()
big = torch.empty(256 << 20, dtype=torch.float32, device="cuda")
for i in range(8):
    Im.dog_pick(v, [3, 5])
    torch.cuda.synchronize()
    if i >= 2:
        Dm.sigmoid_tomo_decode(logits, kernel=3, K=900)
    if i >= 5:
        big.fill_(float(i))
    torch.cuda.synchronize()
print("done")
