"""stem forward: tile 8x4x8 (two z-planes per wave) against tile 8x4x4 - the convolution output must be bit-identical
(same products, same order per output voxel); the BatchNorm partial sums differ in their f32 grouping only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cet_pick_amd import hipops as H
for n, c in ((3, 48), (2, 64), (5, 32), (1, 16)):
    g = torch.Generator().manual_seed(n * 100 + c)
    x = torch.randn(n, c, c, c, 1, generator=g).cuda()
    w = H.conv_weight_param(64, 1, 7); w.data = w.data.cuda(); w.data.normal_(generator=None)
    outs = []
    for z4 in ("", "1"):
        if z4: os.environ["MI_STEM_FWD_Z4"] = "1"
        else: os.environ.pop("MI_STEM_FWD_Z4", None)
        outs.append(H.conv_fwd(x, w, 7, 2, 3).clone())
    torch.cuda.synchronize()
    print(n, c, "bit-identical" if torch.equal(outs[0], outs[1]) else "DIFFER max %.3e" % float((outs[0] - outs[1]).abs().max()))
