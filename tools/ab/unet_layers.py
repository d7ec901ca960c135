"""Per-convolution table of one unet_4 forward on 128 x 512 x 512 (the detector's C3 configuration): shape, GFLOP, time, TFLOP/s.
The shapes are captured by wrapping hipops.conv_fwd / conv_bias_fwd; the times are hipops.PROFILE's HIP-event times of the same launches."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from cet_pick_amd import hipops as H
from cet_pick_amd.models.model import create_model
from cet_pick_amd.synthetic import seeded_state_dict

net = create_model("unet_4", {"hm": 1, "proj": 32}, 32)
net.load_state_dict(seeded_state_dict(net, seed=321))
net = net.cuda().eval()
vol = torch.randn(1, 128, 512, 512, device="cuda")
shapes = []
orig = H.conv_fwd
def wrapped(x, w, k, stride, pad, res=None, relu=False, dil=None, **kw):
    before = len(H.PROFILE) if H.PROFILE is not None else 0
    y = orig(x, w, k, stride, pad, res, relu, dil, **kw)
    if H.PROFILE is not None and len(H.PROFILE) > before:
        shapes.append((tuple(x.shape), w.shape[0], k, stride, dil))
    return y
H.conv_fwd = wrapped
orig_b = H.conv_bias_fwd
def wrapped_b(x, w, bias, k, stride, pad, relu=False, out=None, **kw):     # inference: BatchNorm folded, bias + ReLU epilogue
    before = len(H.PROFILE) if H.PROFILE is not None else 0
    y = orig_b(x, w, bias, k, stride, pad, relu, out, **kw)
    if H.PROFILE is not None and len(H.PROFILE) > before:
        shapes.append((tuple(x.shape), w.shape[0], k, stride, None))
    return y
H.conv_bias_fwd = wrapped_b
with torch.no_grad():
    net(vol); torch.cuda.synchronize()
    H.PROFILE = []
    net(vol); torch.cuda.synchronize()
    prof, H.PROFILE = H.PROFILE, None
agg = {}
for (tag, flops, e0, e1, reps), sh in zip(prof, shapes):
    a = agg.setdefault(sh, [0, 0.0, 0.0]); a[0] += 1; a[1] += flops; a[2] += e0.elapsed_time(e1) / reps
print("%-44s %5s %9s %9s %8s" % ("input shape -> co, k, stride, dil", "calls", "GFLOP", "ms", "TFLOP/s"))
tot = [0.0, 0.0]
for sh, (c, f, ms) in sorted(agg.items(), key=lambda kv: -kv[1][2]):
    print("%-44s %5d %9.1f %9.3f %8.1f" % ("%s -> %d, %s, %d, %s" % sh, c, f / 1e9, ms, f / ms / 1e9))
    tot[0] += f; tot[1] += ms
print("total %.1f GFLOP, %.2f ms, %.1f TFLOP/s (%d profiled launches, %d shapes captured)" % (tot[0] / 1e9, tot[1], tot[0] / tot[1] / 1e9, len(prof), len(shapes)))
