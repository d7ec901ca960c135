"""Timings of the detector-side rows (a12, a22, a23, C5) on one MI355X: loader, unet_4 forward on a full tomogram,
the debiased contrastive loss at N = 12,288 voxels per view, and one semi-supervised training step."""
import json
import os
import sys
import time
from types import SimpleNamespace

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch


def timeit(fn, n=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def latest_traffic(suffix, per_call_key="hbm_bytes_per_step"):
    """(HBM bytes per step / call, note) from the newest profiles/rNN_<suffix> whose kernel-source hash still matches the tree
    (tools/pmc_run.sh + tools/pmc_summary.py: separate --pmc passes, FETCH_SIZE doubled on gfx950), else (None, why)."""
    import glob
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    c = sorted(glob.glob(os.path.join(root, "profiles", "r[0-9][0-9]_" + suffix)))
    if not c:
        return None, "no profiles/rNN_%s" % suffix
    from cet_pick_amd.build import source_sha16
    d = json.load(open(c[-1]))
    if d.get("source_sha16") != source_sha16(d.get("source_prefixes")):
        return None, "stale: profiles/%s was measured on other kernel sources" % os.path.basename(c[-1])
    return d.get(per_call_key), "PMC FETCH_SIZE (x2, gfx950) + WRITE_SIZE over every launch of one call, profiles/%s" % os.path.basename(c[-1])


def run(small=False, only_semi=False, only_unet=False):
    from cet_pick_amd.utils import loader
    from cet_pick_amd.models.model import create_model
    from cet_pick_amd.models.loss import UnbiasedConLoss
    from cet_pick_amd.synthetic import seeded_state_dict
    from cet_pick_amd import hipops as H
    out = {}
    heads = {"hm": 1, "proj": 32}
    g = torch.Generator().manual_seed(0)
    if only_semi:                     # (rocprofv3 of the C5 step alone: profiles/rNN_c5_kernel_stats.csv)
        return _semi_step(out, small, heads, g, create_model, seeded_state_dict)
    if only_unet:                     # (PMC traffic pass, profiles/rNN_unet_traffic.json: exactly five forwards, nothing else)
        net = create_model("unet_4", heads, 32)
        net.load_state_dict(seeded_state_dict(net, seed=321))
        net = net.cuda().eval()
        vol = torch.randn((1, 16, 128, 128) if small else (1, 128, 512, 512), device="cuda")
        with torch.no_grad():
            t = timeit(lambda: net(vol), n=4, warm=1)
        return {"unet4_forward": {"input": list(vol.shape), "ms": t * 1e3, "forwards": 5}}
    # a12: load_rec (xzy order, compress) + preprocess of a 256 x 512 x 512 tomogram already on the host
    shape = (64, 128, 128) if small else (512, 256, 512)          # file order (x, z, y): 256 slices of 512 x 512
    rec = np.random.default_rng(0).standard_normal(shape).astype(np.float32)
    dev_rec = loader.rec_to_device(rec, "xzy", False)
    t = timeit(lambda: loader.preprocess(loader.zscore(dev_rec), 0))
    out["loader_zscore_preprocess"] = {"voxels": int(dev_rec.numel()), "ms": t * 1e3, "voxels_per_sec": dev_rec.numel() / t,
                                       "note": "device-resident volume; z-score + stats + quantise/min-max = 4 passes"}
    # a22: unet_4 forward, 128 x 512 x 512 (3.4 TFLOP)
    heads = {"hm": 1, "proj": 32}
    net = create_model("unet_4", heads, 32)
    net.load_state_dict(seeded_state_dict(net, seed=321))
    net = net.cuda().eval()
    vol = torch.randn((1, 16, 128, 128) if small else (1, 128, 512, 512), device="cuda")
    # the forward's own high-water mark: what the process already holds (inside bench.py: the MoCo engine, the picker's
    # workspace, ...) is subtracted, and the counter is reset here - it is a process-wide maximum otherwise
    torch.cuda.synchronize()
    mem_base = torch.cuda.memory_allocated()
    torch.cuda.reset_peak_memory_stats()
    with torch.no_grad():
        t = timeit(lambda: net(vol), n=3, warm=1)
        H.PROFILE = []                                      # conv launches of one forward: flops and HIP-event times
        net(vol)
        torch.cuda.synchronize()
        prof, H.PROFILE = H.PROFILE, None
    cflops = sum(p[1] for p in prof)
    cms = sum(p[2].elapsed_time(p[3]) / p[4] for p in prof)
    out["unet4_forward"] = {"input": list(vol.shape), "ms": t * 1e3, "input_voxels_per_sec": vol.numel() / t,
                            "conv_gflop": cflops / 1e9, "conv_ms": cms, "conv_tflops": cflops / cms / 1e9,
                            "peak_mem_gb": (torch.cuda.max_memory_allocated() - mem_base) / 2 ** 30,
                            # the convolutions run in the bf16x3 arithmetic: ceiling 2500 / 6 TFLOP/s of f32-equivalent work
                            "roofline": {"bound": "mfma", "kernel": "every convolution launch of one forward (conv_d32 / conv_igemm / conv_smallk / stem2d)",
                                         "achieved": cflops / cms / 1e9, "peak": 2500.0 / 6, "unit": "TFLOP/s",
                                         "frac": cflops / cms / 1e9 / (2500.0 / 6), "traffic": None,
                                         "whole_forward_tflops": cflops / t / 1e12}}
    if not small:
        tr, note = latest_traffic("unet_traffic.json")
        out["unet4_forward"]["roofline"]["traffic"] = tr
        out["unet4_forward"]["roofline"]["traffic_note"] = note + " (bytes per forward of the 128 x 512 x 512 tomogram)"
    # a23: debiased contrastive loss, N = 12,288 per view, dim 32, forward + backward
    n, dim = (2048, 32) if small else (12288, 32)
    g = torch.Generator().manual_seed(0)
    f = torch.nn.functional.normalize(torch.randn(n, dim, generator=g), dim=1).cuda().requires_grad_()
    f2 = torch.nn.functional.normalize(torch.randn(n, dim, generator=g), dim=1).cuda().requires_grad_()
    lab = torch.full((n,), -1.0)
    r = torch.rand(n, generator=g)
    lab[r < 0.3] = 0.0
    lab[r > 0.97] = 1.0
    lab = lab.cuda()
    p1, p2 = torch.rand(n, generator=g).cuda(), torch.rand(n, generator=g).cuda()
    crit = UnbiasedConLoss(0.07, 0.1)
    opt = SimpleNamespace(thresh=0.5, device=torch.device("cuda"))

    def ucl():
        f.grad = None
        sup, unsup = crit(lab, p1, p2, f, f2, opt)
        (sup + 0.1 * unsup).backward()
    t = timeit(ucl)
    flops = 2.0 * (2 * n) ** 2 * dim * (1 + 4)             # S tiles: 1 forward pass + 2 x (S + W.F) backward passes
    out["unbiased_con_loss_fwd_bwd"] = {"N": n, "dim": dim, "ms": t * 1e3, "dense_matrix_bytes_avoided": 4 * (2 * n) ** 2,
                                        "mfma_tflops": flops / t / 1e12}
    return _semi_step(out, small, heads, g, create_model, seeded_state_dict)


def _semi_step(out, small, heads, g, create_model, seeded_state_dict):
    # C5: one semi-supervised training step, 16 pairs of 6 x 64 x 64 crops
    from cet_pick_amd.trains.train_factory import train_factory
    topt = SimpleNamespace(task="semi", arch="unet_4", pn=False, ge=False, tau=0.1, temp=0.07, thresh=0.5, cr_weight=0.1,
                           num_stacks=1, contrastive=True, device=torch.device("cuda"), num_iters=-1, print_iter=0,
                           hide_data_time=True, exp_id="bench", lr=1e-3, hipgraph=False)
    model = create_model("unet_4", heads, 32)
    model.load_state_dict(seeded_state_dict(model, seed=323))
    trainer = train_factory["semi"](topt, model, torch.optim.Adam(model.parameters(), lr=1e-3))
    trainer.set_device([0], None, "cuda")
    b = 4 if small else 16
    x = torch.randn(b, 6, 64, 64, generator=g)
    gt = torch.full((b, 1, 6, 32, 32), -1.0)
    rr = torch.rand(gt.shape, generator=g)
    gt[rr < 0.3] = 0.0
    gt[rr > 0.97] = 1.0
    batch = {"input": x.cuda(), "input_aug": x.flip(-1).cuda(), "hm": gt.cuda(), "flip_prob": 0.2, "meta": {}}
    t = timeit(lambda: trainer.train(1, [dict(batch)]), n=8, warm=3)
    # the step is the debiased contrastive loss: N = b*6*32*32 voxels per view, S = F F^T over 2N rows is formed on the
    # f32 matrix cores once in the forward and twice in each of the two backward kernels, for the supervised and the
    # unsupervised term (models/loss.py): 2 terms x 5 passes x 2 (2N)^2 dim FLOP; the U-Net is noise next to it
    # (round 6: the merged backward executes 3 of those 5 products - S once in the forward, S and W.F once in the backward; the
    # algorithmic count of the two-kernel form is kept so that the rates of the rounds stay comparable)
    n_vox = b * 6 * 32 * 32
    ucl_flop = 2 * 5 * 2.0 * (2 * n_vox) ** 2 * 32
    out["semi_train_step"] = {"pairs": b, "crop": [6, 64, 64], "ms": t * 1e3, "crops_per_sec": 2 * b / t,
                              "voxels_per_view": n_vox, "ucl_gflop": ucl_flop / 1e9,
                              "roofline": {"bound": "mfma", "kernel": "ucl_fwd_kernel + ucl_bwd_kernel<32, 3> (round 6: the backward forms both terms of a "
                                                                      "similarity tile from ONE product with one exponential, tiles double-buffered)",
                                           "achieved": ucl_flop / t / 1e12, "peak": 2500.0 / 6, "unit": "TFLOP/s",
                                           "frac": ucl_flop / t / 1e12 / (2500.0 / 6), "traffic": (None if small else latest_traffic("c5_traffic.json")[0]),
                                           "frac_f32_mfma_peak": ucl_flop / t / 1e12 / 157.3,
                                           "traffic_note": latest_traffic("c5_traffic.json")[1] + " (bytes per training step)",
                                           "note": "whole step time against the loss kernels' FLOPs (they are ~95 % of it); with the "
                                                   "products on the bf16 pipe the kernels are bound by their vector work (one exp and "
                                                   "~10 instructions per similarity), not by the matrix cores; not launch-bound: a "
                                                   "hipGraph of the step would not change it"}}
    return out


if __name__ == "__main__":
    print(json.dumps(run(small="--small" in sys.argv, only_semi="--only-semi" in sys.argv, only_unet="--only-unet" in sys.argv)))
