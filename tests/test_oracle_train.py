"""The training oracle (plain torch CPU) pinned against vectors produced by the reference's own
modules (tests/golden/gen_golden.py).  CPU only."""
import numpy as np
import torch

from oracle import train_ref as T
from cet_pick_amd.synthetic import seeded_state_dict


def _ref_shapes(golden_keys):
    import json, os
    here = os.path.dirname(os.path.abspath(__file__))
    return json.load(open(os.path.join(here, "golden", "ckpt_keys.json")))["moco3d_encoder"]


class _Shape:
    def __init__(self, shapes):
        self._sd = {k: torch.zeros(v) if not k.endswith("num_batches_tracked") else torch.zeros((), dtype=torch.long)
                    for k, v in shapes.items()}

    def state_dict(self):
        return self._sd


def seeded_sd(seed=317):
    sd = seeded_state_dict(_Shape(_ref_shapes(None)), seed=seed)
    # 'proj' and 'pred' are ONE module in the reference (moco_encoder_3d.py:195-236), so
    # load_state_dict leaves the values loaded last - the 'pred.*' entries - in both
    for k in list(sd):
        if k.startswith("pred."):
            sd["proj." + k[5:]] = sd[k]
    return sd


def test_encoder_forward_backward_matches_reference(golden):
    g = golden("enc3d.npz")
    sd = seeded_sd()
    x = torch.randn(4, 1, 32, 32, 32, generator=torch.Generator().manual_seed(99))
    names = T.param_names(sd)
    for n in names:
        sd[n].requires_grad_(True)
    for n in names:
        if n.startswith("proj."):
            sd["pred" + n[4:]] = sd[n]
    acts = {}
    out = T.encoder_forward(sd, x, True, acts)
    np.testing.assert_allclose(out.detach().numpy(), g["proj_train"], rtol=1e-4, atol=1e-5)
    idx = g["sample_idx"]
    for k, v in acts.items():
        f = v.detach().reshape(-1).numpy()
        np.testing.assert_allclose(f[idx % f.size], g[f"act_{k}_sample"], rtol=1e-4, atol=1e-5)
    loss = (out * torch.linspace(-1, 1, 128)[None]).sum() + (out ** 2).sum() * 0.1
    grads = torch.autograd.grad(loss, [sd[n] for n in names])
    for n, gr in zip(names, grads):
        gf = gr.reshape(-1).numpy()
        ref_norm = float(g[f"grad_{n}_norm"])
        assert abs(np.linalg.norm(gf.astype(np.float64)) - ref_norm) <= 1e-3 * ref_norm + 1e-7, n
    np.testing.assert_allclose(sd["bn1.running_mean"].numpy(), g["bn1_running_mean"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(sd["bn1.running_var"].numpy(), g["bn1_running_var"], rtol=1e-5, atol=1e-6)
    sd2 = seeded_sd()
    with torch.no_grad():
        ev = T.encoder_forward(sd2, x, False)
    np.testing.assert_allclose(ev.numpy(), g["proj_eval"], rtol=1e-4, atol=1e-5)


def test_moco_three_steps_match_reference(golden):
    g = golden("moco_3steps.npz")
    sd = seeded_sd()
    ref = T.MocoRef(sd, torch.from_numpy(g["queue0"]), m=0.99, T=0.1, lr=0.05)
    gen = torch.Generator().manual_seed(123)
    torch.randn(128, 64, generator=gen)        # the generator already produced queue0
    B = 8
    for step in range(3):
        im_q = torch.randn(B, 1, 32, 32, 32, generator=gen)
        im_k = im_q.flip(4) + 0.1 * torch.randn(B, 1, 32, 32, 32, generator=gen)
        logits, loss, _ = ref.step(im_q, im_k)
        # step 0 pins the formulas tightly; later steps see fp32 noise amplified through SGD (lr 0.05)
        # and batch-8 BatchNorm, so they pin the state updates (EMA / SGD / enqueue) more loosely
        tol = 1e-4 if step == 0 else 1e-2
        np.testing.assert_allclose(logits.numpy(), g[f"logits_{step}"], rtol=tol, atol=tol)
        assert abs(loss - float(g[f"loss_{step}"])) < tol
        assert ref.ptr == int(g[f"ptr_{step}"])
    np.testing.assert_allclose(ref.queue.numpy(), g["queue_final"], rtol=0, atol=1e-3)
    np.testing.assert_allclose(ref.q["fc.weight"].numpy(), g["q_fc_weight"], rtol=0, atol=1e-3)
    np.testing.assert_allclose(ref.k["fc.weight"].numpy(), g["k_fc_weight"], rtol=0, atol=1e-3)


def test_lr_schedule(golden):
    rows = golden("lr_sched.npz")["rows"]
    for cosine, ep, lr in rows:
        got = T.adjust_learning_rate(0.02, int(ep), [90, 120], 0.1, bool(cosine), 140)
        assert abs(got - lr) < 1e-12
