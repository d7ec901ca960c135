import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from test_oracle_train import seeded_sd
from test_train_gpu import _seeded_encoder
from oracle import train_ref as T
from cet_pick_amd.models.moco import MoCo
from cet_pick_amd import hipops as H
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests/golden/moco_3steps.npz"))
q, k = _seeded_encoder(), _seeded_encoder()
moco = MoCo(q, k, dim=128, r=64, m=0.99, T=0.1).cuda()
moco.queue.copy_(torch.from_numpy(g["queue0"]).cuda())
aq, ak = moco.flatten_parameters()
ref = T.MocoRef(seeded_sd(), torch.from_numpy(g["queue0"]), m=0.99, T=0.1, lr=0.05)
gen = torch.Generator().manual_seed(123); torch.randn(128, 64, generator=gen)
B = 8
moco.train()
for step in range(3):
    im_q = torch.randn(B, 1, 32, 32, 32, generator=gen)
    im_k = im_q.flip(4) + 0.1 * torch.randn(B, 1, 32, 32, 32, generator=gen)
    aq.zero_grad()
    logits, labels = moco(im_q.cuda(), im_k.cuda())
    loss = H.cross_entropy_label0(logits); loss.backward()
    lr_, loss_r, grads = ref.step(im_q, im_k)
    print("step", step, "logits maxdiff vs oracle", float((logits.detach().cpu() - lr_).abs().max()),
          "vs golden", float(np.abs(logits.detach().cpu().numpy() - g[f"logits_{step}"]).max()),
          "oracle vs golden", float(np.abs(lr_.numpy() - g[f"logits_{step}"]).max()))
    worst = []
    for n, p in moco.encoder_q.named_parameters():
        if n not in grads: continue
        a = p.grad.detach().cpu().contiguous(); b = grads[n]
        rel = float((a - b).norm() / (b.norm() + 1e-12))
        worst.append((rel, n, float(b.norm())))
    worst.sort(reverse=True)
    print("  worst grad rel err:", [(round(r, 6), n, round(nb, 4)) for r, n, nb in worst[:6]])
    H.sgd_step_(aq.flat, aq.flat_grad, 0.05)
    # sync the oracle to the GPU state so each step is compared from identical parameters
    for n, p in moco.encoder_q.named_parameters():
        ref.q[n] = p.detach().cpu().contiguous().clone()
    for n, p in moco.encoder_k.named_parameters():
        ref.k[n] = p.detach().cpu().contiguous().clone()
    for n, bfr in moco.encoder_q.named_buffers():
        ref.q[n] = bfr.detach().cpu().clone()
    for n, bfr in moco.encoder_k.named_buffers():
        ref.k[n] = bfr.detach().cpu().clone()
    print("  queue maxdiff", float((moco.queue.cpu() - ref.queue).abs().max()))
    ref.queue = moco.queue.cpu().clone()
