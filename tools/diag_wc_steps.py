"""Error levels of the well-conditioned three-step MoCo run (tests/golden/moco_3steps_wc.npz) on the GPU: per step, the
relative L2 error of every sampled gradient against the reference's, and the norm errors.  Evidence for the tolerances of
tests/test_train_gpu.py::test_moco_three_wellconditioned_steps_match_reference."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_train_gpu import _seeded_encoder
from cet_pick_amd.models.moco import MoCo
from cet_pick_amd import hipops as H

g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "moco_3steps_wc.npz"))
torch.manual_seed(7)
moco = MoCo(_seeded_encoder(), _seeded_encoder(), dim=128, r=64, m=0.99, T=0.1).cuda()
moco.queue.copy_(torch.from_numpy(g["queue0"]).cuda())
aq, ak = moco.flatten_parameters()
gen = torch.Generator().manual_seed(123)
torch.randn(128, 64, generator=gen)
idx = g["sample_idx"]
moco.train()
for step in range(3):
    im_q = torch.randn(8, 1, 32, 32, 32, generator=gen)
    im_k = im_q.flip(4) + 0.1 * torch.randn(8, 1, 32, 32, 32, generator=gen)
    aq.zero_grad()
    logits, labels = moco(im_q.cuda(), im_k.cuda())
    loss = H.cross_entropy_label0(logits)
    loss.backward()
    print("step", step, "logit err", np.abs(logits.detach().cpu().numpy() - g[f"logits_{step}"]).max(), "loss", float(loss), float(g[f"loss_{step}"]))
    worst = 0
    for n, p in moco.encoder_q.named_parameters():
        if f"gnorm_{step}_{n}" not in g.files:
            continue
        gf = p.grad.detach().cpu().contiguous().reshape(-1).numpy()
        want = float(g[f"gnorm_{step}_{n}"])
        ne = abs(np.linalg.norm(gf.astype(np.float64)) - want) / max(want, 1e-30)
        if want > 1e-4:
            worst = max(worst, ne)
        if f"gsample_{step}_{n}" in g.files:
            ws = g[f"gsample_{step}_{n}"]
            d = gf[idx % gf.size] - ws
            print("   %-32s norm_err %.2e  sample relL2 %.2e  max|d|/max|ws| %.2e" % (n, ne, np.linalg.norm(d) / np.linalg.norm(ws), np.abs(d).max() / np.abs(ws).max()))
    print("   worst norm err", worst)
    H.sgd_step_(aq.flat, aq.flat_grad, 1e-3)
