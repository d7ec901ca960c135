"""In-kernel time stamps of decode1_kernel (tools/var/lib_stamps.so: s_memrealtime, 100 MHz, lane 0 of every workgroup):
where the one-launch decode spends its time.  CETPICK_HIP_LIB=tools/var/lib_stamps.so python tools/ab/decode1_stamps.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from cet_pick_amd import _lib as L
from cet_pick_amd.synthetic import make_logits
from cet_pick_amd.models import decode as Dm

shape = (128, 256, 256)
logits = torch.as_tensor(make_logits(shape, seed=317)).cuda()[None, None]
for _ in range(5):
    Dm.sigmoid_tomo_decode(logits, kernel=3, K=900)
torch.cuda.synchronize()
d, h, w = shape
ws = Dm._decode_workspace(L.lib().mi_decode_workspace_bytes(d, h, w, 900), logits.device)
n = d * h * w
n_wg = 256
off = 8448 + n * 8 - n_wg * 16 * 8
st = ws[off:off + n_wg * 16 * 8].view(torch.int64).view(n_wg, 16).cpu().numpy().astype(np.float64) / 100.0   # us
t0 = st[:, 0].min()
names = ["entry", "march end", "hist+segments done", "keep staged", "before ticket", "after ticket"]
for k, nm in enumerate(names):
    c = st[:, k] - t0
    print("%-20s min %7.2f  median %7.2f  max %7.2f us" % (nm, c.min(), np.median(c), c.max()))
last = int(np.argmax(st[:, 6] > 0)) if (st[:, 6] > 0).any() else -1
for w_ in range(n_wg):
    if st[w_, 8] > st[w_, 0]:
        last = w_
order = [(5, "after ticket"), (9, "table loads issued"), (10, "bounds loaded"), (11, "table arrived, histogram adds issued"), (6, "histogram complete (barrier)"),
         (12, "threshold known"), (7, "survivors in LDS"), (13, "bucket counts"), (14, "bucket starts"), (8, "ranked + emitted")]
print("last workgroup %d:" % last)
for k, nm in order:
    print("   %-40s %7.2f us" % (nm, st[last, k] - t0))
