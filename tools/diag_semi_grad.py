"""Where does the semi-supervised step's gradient error come from?  (a) the loss alone: GPU loss module fed the CPU
oracle's outputs as leaves - gradients w.r.t. hm / proj logits against float64; (b) per-parameter errors of the whole step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from types import SimpleNamespace
import numpy as np, torch
from oracle import loss_ref as OL, unet_ref as OU
from cet_pick_amd.models.model import create_model
from cet_pick_amd.synthetic import seeded_state_dict
from cet_pick_amd.trains.tomo_cr_semi_trainer import TomoCRSemiLoss

heads = {"hm": 1, "proj": 32}
opt = SimpleNamespace(task="semi", arch="unet_4", pn=False, ge=False, tau=0.1, temp=0.07, thresh=0.5, cr_weight=0.1,
                      num_stacks=1, contrastive=True, device=torch.device("cuda"))
model = create_model(opt.arch, heads, 32)
sd0 = seeded_state_dict(model, seed=323)
for k in ("hm.weight", "proj.weight"):
    sd0[k] = sd0[k] * 0.3
g = torch.Generator().manual_seed(2)
b, d, h, w = 2, 4, 48, 48
x = torch.randn(b, d, h, w, generator=g)
x_aug = x.flip(-1) + 0.05 * torch.randn(b, d, h, w, generator=g)
gt = torch.full((b, 1, d, h // 2, w // 2), -1.0)
r = torch.rand(gt.shape, generator=g)
gt[r < 0.3] = 0.0
gt[(r >= 0.3) & (r < 0.4)] = 0.6
gt[r > 0.96] = 1.0


def outs(dt):
    rsd = {k: (v.to(dt) if v.is_floating_point() else v.clone()) for k, v in sd0.items()}
    with torch.no_grad():
        o1 = OU.tomo_conv_unet_forward(rsd, x.to(dt), 4, heads, training=True)
        o2 = OU.tomo_conv_unet_forward(rsd, x_aug.to(dt), 4, heads, training=True)
    return [t.detach().clone().requires_grad_(True) for t in (o1["hm"], o2["hm"], o1["proj"], o2["proj"])]


def cpu_loss_grads(dt):
    leaves = outs(dt)
    res = OL.tomo_cr_semi_loss(leaves[0], leaves[1], leaves[2], leaves[3], gt.to(dt), 0.2, opt.tau, opt.temp, opt.thresh, opt.cr_weight)
    terms = {}
    for name, t in zip(("loss", "hm_loss", "cr_loss", "consis_loss"), res):
        gr = torch.autograd.grad(t, leaves, retain_graph=True, allow_unused=True)
        terms[name] = [None if q is None else q.double() for q in gr]
    return terms, [float(t) for t in res]


t32, v32 = cpu_loss_grads(torch.float32)
t64, v64 = cpu_loss_grads(torch.float64)
# GPU loss module on the fp32 oracle outputs (logits of hm are pre-sigmoid in both)
leaves = [t.detach().float().cuda().requires_grad_(True) for t in outs(torch.float32)]
crit = TomoCRSemiLoss(opt)
o = [{"hm": leaves[0] * 1.0, "proj": leaves[2] * 1.0}]
ocr = [{"hm": leaves[1] * 1.0, "proj": leaves[3] * 1.0}]
loss, stats = crit(o, {"hm": gt.cuda(), "flip_prob": 0.2}, 1, "train", output_cr=ocr)
print("loss values  gpu %s" % {k: float(v) for k, v in stats.items()})
print("             f32 %s\n             f64 %s" % (v32, v64))
for name in ("loss", "hm_loss", "cr_loss", "consis_loss"):
    gr = torch.autograd.grad(stats[name], leaves, retain_graph=True, allow_unused=True)
    for i, lab in enumerate(("d/dhm1", "d/dhm2", "d/dproj1", "d/dproj2")):
        r64 = t64[name][i]
        if r64 is None or gr[i] is None:
            continue
        sc = float(r64.norm()) + 1e-30
        print("%-12s %-9s |g64| %.3e   gpu err %.3e   cpu32 err %.3e" % (name, lab, sc, float((gr[i].cpu().double() - r64).norm()) / sc,
                                                                      float((t32[name][i] - r64).norm()) / sc))

# (b) the whole step: per-parameter gradient errors against float64
from cet_pick_amd.trains.train_factory import train_factory
topt = SimpleNamespace(**opt.__dict__, num_iters=-1, print_iter=0, hide_data_time=True, exp_id="t", lr=1e-3, hipgraph=False)
model.load_state_dict(sd0)
trainer = train_factory["semi"](topt, model, torch.optim.SGD(model.parameters(), lr=1e-3))
trainer.set_device([0], None, "cuda")


def cpu_step(dt):
    rsd = {k: (v.to(dt) if v.is_floating_point() else v.clone()).clone().requires_grad_(
        v.is_floating_point() and not k.endswith(("running_mean", "running_var"))) for k, v in sd0.items()}
    o1 = OU.tomo_conv_unet_forward(rsd, x.to(dt), 4, heads, training=True)
    o2 = OU.tomo_conv_unet_forward(rsd, x_aug.to(dt), 4, heads, training=True)
    res = OL.tomo_cr_semi_loss(o1["hm"], o2["hm"], o1["proj"], o2["proj"], gt.to(dt), 0.2, opt.tau, opt.temp, opt.thresh, opt.cr_weight)
    res[0].backward()
    return rsd


s32, s64 = cpu_step(torch.float32), cpu_step(torch.float64)
trainer.train(1, [{"input": x, "input_aug": x_aug, "hm": gt, "flip_prob": 0.2, "meta": {}}])
for name, prm in model.named_parameters():
    g64 = s64[name].grad
    upd = (prm.detach().cpu().double() - sd0[name].double()) / (-1e-3)
    sc = float(g64.norm()) + 1e-30
    print("%-34s |g64| %.3e  gpu err %.3e  cpu32 err %.3e" % (name, sc, float((upd - g64).norm()) / sc, float((s32[name].grad.double() - g64).norm()) / sc))
