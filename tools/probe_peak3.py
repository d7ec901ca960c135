"""Timing probe for the (3,3,3) register march: which part of the kernel costs what, against a plain device copy."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cet_pick_amd import _lib as L
from cet_pick_amd.synthetic import make_logits
from cet_pick_amd.models import decode as Dm


def timeit(fn, n=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


shape = (128, 256, 256) if len(sys.argv) < 4 else tuple(int(v) for v in sys.argv[1:4])
logits = torch.as_tensor(make_logits(shape, seed=317)).cuda()
heat = torch.sigmoid(logits).clamp(1e-4, 1 - 1e-4)
out = torch.empty_like(logits)
d, h, w = shape
print("copy_ %.1f us" % timeit(lambda: out.copy_(logits)))
print("fill_ %.1f us" % timeit(lambda: out.fill_(1.0)))
print("sum   %.1f us" % timeit(lambda: logits.sum()))
lib = L.lib()
ws = L.workspace(lib.mi_decode_workspace_bytes(d, h, w, 900) * 2, logits.device, "probe")
dets = torch.empty((900, 5), device="cuda")
for minz, dbg in ((8, 0), (4, 0), (2, 0), (16, 0)):
    os.environ["MI_PEAK3_MINZ"] = str(minz)
    os.environ["MI_PEAK3_WAVES"] = "8192"
    t_nms = timeit(lambda: L.check(lib.mi_nms3d(L.ptr(heat), L.ptr(out), d, h, w, 3, 3, L.stream()), "nms"))
    t_dec = timeit(lambda: L.check(lib.mi_sigmoid_nms_topk(L.ptr(heat), None, d, h, w, 3, 0, 0, 900, L.ptr(dets), None,
                                                           L.ptr(ws), ws.numel(), L.stream()), "dec"))
    t_sig = timeit(lambda: L.check(lib.mi_sigmoid_nms_topk(L.ptr(logits), L.ptr(out), d, h, w, 3, 0, 1, 900, L.ptr(dets), None,
                                                           L.ptr(ws), ws.numel(), L.stream()), "dec"))
    print("minz %d dbg %d: nms3d(dense out) %.1f us | decode no-sigmoid (cands only) %.1f us | fused sigmoid decode %.1f us"
          % (minz, dbg, t_nms, t_dec, t_sig))
