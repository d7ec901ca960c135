"""The SimSiam-2D train loop at the reference's documented exploration configuration (docs/explore.md:67:
`simsiam_main.py simsiam3d --arch simsiam2d_18 --bbox 36 --batch_size 256 --lr 1e-3`; model
models/networks/simsiam_model_2d.py:617-819, loop simsiam_main.py:25-166, trainer trains/base_trainer.py:446-552) on one
MI355X: the C2 tomogram (128 x 512 x 512) written as an MRC file -> TomoFileSimSiamDataset (device load_rec / preprocess,
DoG picks, 3 x 36 x 36 crops summed over z) -> `trainer.train()`.

    python tools/bench_simsiam2d.py [--small] [--no-cpu] [--only-step]

Record: ms per step (one batch of 256 crop PAIRS: two views forward, backward, SGD), crop pairs/s, the conv family's MFMA
roofline (algorithmic 2 views x 3 x 1.063 GFLOP per pair), the CPU oracle on the host cores, the entry point's own rate.
`--only-step`: the resident-batch step loop alone (the rocprofv3 target: profiles/rNN_simsiam2d_kernel_stats.csv).
"""
import json
import os
import shutil
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

GFLOP_FWD_PER_VIEW = 1.063          # 36 x 36 view through the trunk (stride-1 stem keeps 36^2 at 64 channels)
PEAK_BF16X3 = 2500.0 / 6.0


def _bbox():
    return int(sys.argv[sys.argv.index("--bbox") + 1]) if "--bbox" in sys.argv else 36


def _build(batch, shape, tmp):
    from cet_pick_amd import simsiam_main
    from cet_pick_amd.opts import opts
    from cet_pick_amd.synthetic import make_tomo
    from cet_pick_amd.utils import mrc
    os.makedirs(os.path.join(tmp, "data"), exist_ok=True)
    vol, _ = make_tomo(shape, seed=317)
    mrc.write(os.path.join(tmp, "data", "c2.rec"), vol)
    with open(os.path.join(tmp, "data", "train_images.txt"), "w") as f:
        f.write("image_name\trec_path\nc2\tc2.rec\n")
    opt = opts().parse(["simsiam3d", "--arch", "simsiam2d_18", "--dataset", "simsiam3d", "--order", "zxy", "--bbox", str(_bbox()),
                        "--batch_size", str(batch), "--lr", "0.001", "--exp_id", "bench_simsiam2d", "--debug", "0", "--dog", "3,5",
                        "--num_epochs", "1"])
    return simsiam_main.build(opt)


def cpu_baseline(batch, seed=317):
    """oracle/train_ref.py (torch fp32, all host cores): two-view forward, loss, backward, SGD on `batch` pairs, once after a
    warm-up - the bounded sample of the batch-256 step (the per-pair cost of the CPU path does not depend on the batch)."""
    from oracle import train_ref as T
    from cet_pick_amd.models.networks.simsiam_model_2d import get_simsiam2d_net_small
    from cet_pick_amd.synthetic import seeded_state_dict
    try:
        ncpu = len(os.sched_getaffinity(0))
    except AttributeError:
        ncpu = os.cpu_count() or 1
    torch.set_num_threads(max(1, min(ncpu, 16)))
    net = get_simsiam2d_net_small(18, {"proj": 128, "pred": 128}, 128)
    sd = {k: v.detach().clone().contiguous() for k, v in seeded_state_dict(net, seed=seed).items()}
    names = [k for k in sd if k.endswith((".weight", ".bias"))]
    g = torch.Generator().manual_seed(seed)
    x1 = torch.randn(batch, 1, 36, 36, generator=g)
    x2 = x1.flip(-1)

    def step():
        for n in names:
            sd[n].requires_grad_(True)
        p1, z1, p2, z2 = T.simsiam_forward(sd, x1, x2, True)
        loss, _ = T.simsiam_loss(p1, z1, p2, z2)
        grads = torch.autograd.grad(loss, [sd[n] for n in names], allow_unused=True)
        with torch.no_grad():
            for n, gr in zip(names, grads):
                sd[n] = (sd[n] - 1e-3 * gr).detach() if gr is not None else sd[n].detach()
    step()
    t0 = time.perf_counter()
    step()
    dt = time.perf_counter() - t0
    return batch / dt, torch.get_num_threads()


def run(small=False, with_cpu=True, only_step=False, steps=20):
    from cet_pick_amd import hipops as H
    batch = 32 if small else 256
    shape = (48, 192, 192) if small else (128, 512, 512)
    cwd = os.getcwd()
    tmp = tempfile.mkdtemp(prefix="cetpick_simsiam2d_")
    try:
        os.chdir(tmp)
        opt, model, optimizer, trainer, dataset = _build(batch, shape, tmp)
        dev = opt.device
        if "--no-graph" in sys.argv and getattr(trainer, "engine", None) is not None:      # (PMC passes: every launch a plain dispatch)
            trainer.engine.use_graph = False
        # ---- the step alone, on ONE resident batch (inputs in HBM before the timed region) ----
        dataset.set_epoch(0)
        first = next(iter(dataset))
        x1, x2 = first["input"].contiguous(), first["input_aug"].contiguous()

        def step():
            return trainer.train_step(x1, x2)
        for _ in range(4):                               # sizes workspaces; an engine captures its hipGraph on the third call
            step()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for _ in range(steps):
            loss = step()
        e1.record()
        torch.cuda.synchronize()
        wall_ms = (time.perf_counter() - t0) / steps * 1e3
        ev_ms = e0.elapsed_time(e1) / steps
        flop_step = 2 * 3 * GFLOP_FWD_PER_VIEW * 1e9 * batch * (_bbox() / 36.0) ** 2
        rec = {"workload": "simsiam_main.py simsiam3d --arch simsiam2d_18 --bbox %d --batch_size %d --lr 1e-3 (docs/explore.md:67): "
                           "%d pairs of %d x %d views per step, crops of the %s tomogram" % (_bbox(), batch, batch, _bbox(), _bbox(), "x".join(map(str, shape))),
               "ms": wall_ms, "ms_hip_events": ev_ms, "crop_pairs_per_sec": batch / wall_ms * 1e3, "batch": batch,
               "crops_in_dataset": int(dataset.num_samples), "final_loss": float(loss),
               "engine": type(getattr(trainer, "engine", None)).__name__ if getattr(trainer, "engine", None) is not None else None,
               "hipgraph": bool(getattr(getattr(trainer, "engine", None), "_graph", None) is not None),
               "step_graph_nodes": (trainer.engine.node_counts() if getattr(trainer, "engine", None) is not None
                                    and hasattr(trainer.engine, "node_counts") else None),
               "step_mfma_frac_of_peak": flop_step / (wall_ms * 1e-3) / 1e12 / PEAK_BF16X3}
        if only_step:
            return rec
        # ---- the conv family by itself: every conv call of one eager step timed between two HIP events ----
        H.PROFILE = []
        trainer.train_step(x1, x2, eager=True)
        torch.cuda.synchronize()
        prof, H.PROFILE = H.PROFILE, None
        cflop = sum(p[1] for p in prof)
        cms = sum(p[2].elapsed_time(p[3]) / p[4] for p in prof)
        by = {}
        for tag, f, a, b, r in prof:
            d = by.setdefault(tag, [0, 0.0, 0.0])
            d[0] += 1; d[1] += f; d[2] += a.elapsed_time(b) / r
        rec["roofline"] = {"bound": "mfma", "kernel": "the conv family of one step: every 2-D convolution forward / data gradient / "
                                                        "weight gradient launch (+ its split-K reduce) of both views",
                           "achieved": cflop / cms / 1e9, "peak": PEAK_BF16X3, "unit": "TFLOP/s", "frac": cflop / cms / 1e9 / PEAK_BF16X3,
                           "traffic": None, "launches_per_step": len(prof), "kernel_ms_per_step": cms,
                           "algorithmic_gflop_per_step": cflop / 1e9,
                           "algorithmic_gflop_per_pair": cflop / 1e9 / batch,
                           "by_mode": {t: {"launches": v[0], "gflop": v[1] / 1e9, "ms": v[2], "tflops": v[1] / v[2] / 1e9}
                                       for t, v in sorted(by.items())},
                           "measured": "per conv call of one eager step: 8 back-to-back launches between two HIP events on the "
                                       "launch stream; compare profiles/r06_simsiam2d_kernel_stats.csv"}
        if not small:
            from tools.bench_detector import latest_traffic
            tr, note = latest_traffic("simsiam2d_traffic.json")
            rec["roofline"]["traffic"] = tr
            rec["roofline"]["traffic_note"] = note + " (bytes per training step, conv family)"
        if "--calls" in sys.argv:                        # every conv call of the step in issue order: (mode, GFLOP, us, TFLOP/s)
            rec["calls"] = [(t, round(f / 1e9, 3), round(a.elapsed_time(b) / r * 1e3, 1), round(f / (a.elapsed_time(b) / r) / 1e9, 1))
                            for t, f, a, b, r in prof]
        # ---- the entry point's own rate: trainer.train(epoch, dataset) with its loader and meters ----
        iters = len(dataset)
        dataset.set_epoch(1)
        trainer.train(1, dataset)
        torch.cuda.synchronize()
        epochs = max(1, -(-20 // max(iters, 1)))
        t0 = time.perf_counter()
        for e in range(2, 2 + epochs):
            dataset.set_epoch(e)
            log, _ = trainer.train(e, dataset)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        rec["entry_point"] = {"crop_pairs_per_sec": epochs * iters * batch / dt, "ms_per_iteration": dt / (epochs * iters) * 1e3,
                              "iterations": epochs * iters, "loss": float(log["loss"]),
                              "ratio_to_step": (epochs * iters * batch / dt) / rec["crop_pairs_per_sec"]}
        if hasattr(trainer, "close"):
            trainer.close()
        if with_cpu:
            cb = 16 if small else 64
            v, cores = cpu_baseline(cb)
            rec["cpu_baseline"] = {"value": v, "unit": "crop pairs/sec", "cores": cores, "kind": "port",
                                   "sample": "one step of %d pairs (after one warm-up step) of the same network with "
                                             "oracle/train_ref.py simsiam_forward + simsiam_loss + autograd + SGD (torch fp32, all host cores)" % cb}
        return rec
    finally:
        os.chdir(cwd)
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    print(json.dumps(run(small="--small" in sys.argv, with_cpu="--no-cpu" not in sys.argv, only_step="--only-step" in sys.argv)))
