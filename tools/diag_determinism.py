"""Run-to-run determinism of the MoCo step at the bench shapes: R fresh engines from the same seed, K steps each, eager and
replayed from a hipGraph - every run must end in bit-identical weights, queue and losses."""
import os, sys, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cet_pick_amd.models.networks.moco_encoder_3d import get_moco_net_small_3d
from cet_pick_amd.models.moco import MoCo
from cet_pick_amd.trains.moco_engine import MocoStepEngine

R = int(sys.argv[1]) if len(sys.argv) > 1 else 4
K = int(sys.argv[2]) if len(sys.argv) > 2 else 8
B = int(sys.argv[3]) if len(sys.argv) > 3 else 64


def run(use_graph):
    torch.manual_seed(5)
    heads = {"proj": 256, "pred": 256}
    moco = MoCo(get_moco_net_small_3d(18, heads, 0), get_moco_net_small_3d(18, heads, 0), dim=128, r=4096, m=0.99, T=0.1).cuda()
    moco.train()
    eng = MocoStepEngine(moco, lr=1e-3, use_graph=use_graph)
    g = torch.Generator(device="cuda").manual_seed(11)
    losses = []
    for i in range(K):
        x = torch.randn(B, 1, 32, 32, 32, device="cuda", generator=g)
        losses.append(float(eng.step(x, x.flip(4))))
    torch.cuda.synchronize()
    hsh = hashlib.sha256()
    for t in (eng.arena_q.flat, eng.arena_k.flat, moco.queue):
        hsh.update(t.detach().cpu().numpy().tobytes())
    eng.close()
    return hsh.hexdigest()[:16], losses


for mode in (False, True):
    sigs = [run(mode) for _ in range(R)]
    ok = all(s == sigs[0] for s in sigs)
    print("graph" if mode else "eager", "deterministic" if ok else "NOT DETERMINISTIC", [s[0] for s in sigs], flush=True)
    if not ok:
        for s in sigs:
            print("   ", s[0], ["%.9g" % v for v in s[1]])
