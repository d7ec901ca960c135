"""s2 forward kernel vs the generic launches vs float64 on ill-conditioned inputs (large common mode: cancelling sums)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from cet_pick_amd import hipops as H

def cl(t): return t.permute(0, 2, 3, 4, 1).contiguous().cuda()
for (n, gi, ci, co) in ((8, 8, 64, 128), (8, 4, 128, 256)):
    for offs in (0.0, 1e3, 1e6):
        g = torch.Generator().manual_seed(1)
        w = torch.randn(co, ci, 3, 3, 3, generator=g) * 0.05
        wd = torch.randn(co, ci, 1, 1, 1, generator=g) * 0.05
        p = H.conv_weight_param(co, ci, 3); p.data = p.data.cuda(); p.data.copy_(w.cuda())
        pd = H.conv_weight_param(co, ci, 1); pd.data = pd.data.cuda(); pd.data.copy_(wd.cuda())
        x = torch.relu(torch.randn(n, ci, gi, gi, gi, generator=g) + 0.3) * (1 + offs) + offs
        r64 = (F.conv3d(x.double(), w.double(), stride=2, padding=1).permute(0, 2, 3, 4, 1), F.conv3d(x.double(), wd.double(), stride=2).permute(0, 2, 3, 4, 1))
        r32 = (F.conv3d(x, w, stride=2, padding=1).permute(0, 2, 3, 4, 1), F.conv3d(x, wd, stride=2).permute(0, 2, 3, 4, 1))
        got = H.conv_fwd_s2_block(cl(x), p, pd)
        os.environ["MI_CONV_NO_S2FWD"] = "1"
        gen = (H.conv_fwd(cl(x), p, 3, 2, 1, None, False), H.conv_fwd(cl(x), pd, 1, 2, 0))
        os.environ["MI_CONV_NO_S2FWD"] = "0"
        for i, nm in enumerate(("conv", "shortcut")):
            ref = r64[i]
            pre = ref if i else ref           # compare the pre-activation where possible
            a = got[i].cpu().double(); b = gen[i].cpu().double(); c = r32[i].double()
            if i == 0:
                ref = torch.relu(ref); b = torch.relu(b); c = torch.relu(c)
            sc = float(ref.abs().max())
            print("grid %d offs %g %-8s  max|ref| %.3e   err/max: s2 %.2e  generic %.2e  cpu32 %.2e   rms: s2 %.2e generic %.2e cpu32 %.2e" % (
                gi, offs, nm, sc, float((a - ref).abs().max()) / sc, float((b - ref).abs().max()) / sc, float((c - ref).abs().max()) / sc,
                float((a - ref).pow(2).mean().sqrt()) / sc, float((b - ref).pow(2).mean().sqrt()) / sc, float((c - ref).pow(2).mean().sqrt()) / sc))
