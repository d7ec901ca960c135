"""Diagnostic: per-parameter gradient error (vs the float64 oracle) of three MoCo steps, direct layer1 kernel on / off."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from oracle import train_ref as T
from test_oracle_train import seeded_sd
from test_train_gpu import _seeded_encoder
from cet_pick_amd.models.moco import MoCo
from cet_pick_amd import hipops as H

g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "moco_3steps.npz"))
torch.manual_seed(7)
q, k = _seeded_encoder(), _seeded_encoder()
moco = MoCo(q, k, dim=128, r=64, m=0.99, T=0.1).cuda()
moco.queue.copy_(torch.from_numpy(g["queue0"]).cuda())
aq, ak = moco.flatten_parameters()
ref64 = T.MocoRef({k: (v.double() if v.is_floating_point() else v) for k, v in seeded_sd().items()},
                  torch.from_numpy(g["queue0"]).double(), m=0.99, T=0.1, lr=0.05)
gen = torch.Generator().manual_seed(123)
torch.randn(128, 64, generator=gen)
moco.train()
for step in range(3):
    im_q = torch.randn(8, 1, 32, 32, 32, generator=gen)
    im_k = im_q.flip(4) + 0.1 * torch.randn(8, 1, 32, 32, 32, generator=gen)
    l64, loss64, grads64 = ref64.step(im_q.double(), im_k.double())
    state = {n: t.detach().clone() for n, t in list(moco.named_parameters()) + list(moco.named_buffers())}
    acts, gacts = {}, {}
    def fhook(name, mode):
        def f(mod, inp, out):
            t = out if torch.is_tensor(out) else out[0]
            acts[(name, mode)] = t.detach().clone()
            if t.requires_grad:
                t.register_hook(lambda gr, k=(name, mode): gacts.__setitem__(k, gr.detach().clone()))
        return f
    for mode in ("f32", "1", "0"):
        os.environ["MI_CONV_ARITH"] = "f32" if mode == "f32" else "bf16x3"
        os.environ["MI_CONV_NO_DIRECT"] = "1" if mode == "f32" else mode
        hs = [m.register_forward_hook(fhook(n, mode)) for n, m in moco.encoder_q.named_modules()
              if n and n.count(".") <= 1]
        with torch.no_grad():
            for n, t in list(moco.named_parameters()) + list(moco.named_buffers()):
                t.copy_(state[n])
        aq.zero_grad()
        logits, labels = moco(im_q.cuda(), im_k.cuda())
        loss = H.cross_entropy_label0(logits)
        loss.backward()
        errs = []
        for n, p in moco.encoder_q.named_parameters():
            if n not in grads64: continue
            a = p.grad.detach().cpu().contiguous().double()
            if n != "fc.bias": errs.append((float((a - grads64[n]).norm() / (grads64[n].norm() + 1e-30)), n))
        for hnd in hs: hnd.remove()
        errs.sort(reverse=True)
        print("step", step, "NO_DIRECT=" + mode, "logit err %.2e" % float((logits.detach().cpu().double() - l64).abs().max()),
              " worst:", ["%s %.1e" % (n, e) for e, n in errs[:4]], flush=True)
    for (name, mode) in sorted(acts):
        if mode != "0": continue
        a, b = acts[(name, "0")], acts[(name, "f32")]
        line = "   %-22s act maxdiff %.2e (max %.2e) signflips %d" % (name, float((a - b).abs().max()), float(b.abs().max()), int(((a > 0) != (b > 0)).sum()))
        if (name, "0") in gacts and (name, "f32") in gacts:
            ga, gb = gacts[(name, "0")], gacts[(name, "f32")]
            line += "  | grad reldiff %.2e" % float((ga - gb).norm() / (gb.norm() + 1e-30))
        print(line, flush=True)
    H.sgd_step_(aq.flat, aq.flat_grad, 0.05)
    for enc, dst64 in ((moco.encoder_q, ref64.q), (moco.encoder_k, ref64.k)):
        for n, t in list(enc.named_parameters()) + list(enc.named_buffers()):
            dst64[n] = t.detach().cpu().contiguous().double() if t.is_floating_point() else t.detach().cpu().clone()
    ref64.queue = moco.queue.cpu().double()
