"""Where does the GPU's gradient error at batch 64 come from?  Plain encoder forward / backward (no engine) on seeded weights
against oracle/train_ref.encoder_forward in fp32 and float64: per-parameter relative L2 errors, and per-activation errors."""
import os, sys
import numpy as np
import torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from oracle import train_ref as T
from cet_pick_amd.models.networks.moco_encoder_3d import get_moco_net_small_3d
from cet_pick_amd.synthetic import seeded_state_dict
from cet_pick_amd import hipops as H

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
heads = {"proj": 256, "pred": 256}
enc = get_moco_net_small_3d(18, heads, 0)
sd0 = seeded_state_dict(enc, seed=317)
for kk in [k for k in sd0 if k.startswith("pred.")]:
    sd0["proj." + kk[5:]] = sd0[kk]
enc.load_state_dict(sd0)
enc = enc.cuda().train()
g = torch.Generator().manual_seed(5)
x = torch.randn(B, 1, 32, 32, 32, generator=g)
wv = torch.randn(B, 128, generator=g)
out = enc(x.cuda())[0]["proj"]
loss = (torch.nn.functional.normalize(out, dim=1) * wv.cuda()).sum()
loss.backward()

def ref(dt):
    sd = {k: (v.to(dt) if v.is_floating_point() else v.clone()).clone().requires_grad_(k.endswith(T.PARAM_SUFFIX)) for k, v in sd0.items()}
    o = T.encoder_forward(sd, x.to(dt), True)
    l = (torch.nn.functional.normalize(o, dim=1) * wv.to(dt)).sum()
    l.backward()
    return o.detach(), sd
o32, sd32 = ref(torch.float32)
o64, sd64 = ref(torch.float64)
rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
print("output: gpu %.2e  cpu32 %.2e" % (rel(out.detach().cpu(), o64), rel(o32, o64)))
for n, p in enc.named_parameters():
    if n.startswith("pred.") or sd64[n].grad is None: continue
    a = p.grad.detach().cpu().contiguous()
    print("%-34s gpu %.2e  cpu32 %.2e   |g| %.3e" % (n, rel(a, sd64[n].grad), rel(sd32[n].grad, sd64[n].grad), float(sd64[n].grad.norm())))
