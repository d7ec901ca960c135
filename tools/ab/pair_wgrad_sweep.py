"""pair_wgrad_kernel on one geometry with the segment count of MI_PAIR_WGRAD_SPLITS (run under rocprofv3 --kernel-trace --stats:
tools/gprof.sh): argv = N Di Ci Co k stride reps."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from cet_pick_amd import hipops as H

n, d, ci, co, k, s = (int(v) for v in sys.argv[1:7])
reps = int(sys.argv[7]) if len(sys.argv) > 7 else 20
pad = 1 if k == 3 else 0
do = (d + 2 * pad - k) // s + 1
x = torch.randn(n, d, d, d, ci, device="cuda")
dy = torch.randn(n, do, do, do, co, device="cuda")
w = H.conv_weight_param(co, ci, k); w.data = w.data.cuda()
for _ in range(reps):
    w.grad = None
    H.conv_wgrad_into(x, dy, w, k, s, pad)
torch.cuda.synchronize()
from cet_pick_amd import _lib as L
print(L.lib().mi_debug_last_conv_kernel().decode())
