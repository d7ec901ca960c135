"""Driver benchmark.  `python bench.py --gpus N --steps K --warmup W` (N>1: one rank per GPU under
torch.distributed.run, RCCL).  Prints ONE JSON line on rank 0.

step      = one MoCo-3D training step (BASELINE.json configs[1]): q forward + EMA + k forward +
            InfoNCE + backward + SGD on a batch of 64 pairs of 32^3 sub-tomogram views cut from a
            synthetic 128x512x512 tomogram; inputs are resident in HBM before the timed region.
value     = sub-tomograms/s over all ranks (one PAIR of views counts as one sub-tomogram).
roofline  = conv_igemm_kernel (every conv / linear fwd, dgrad, wgrad launch): algorithmic FLOPs
            2*M*N*K per launch / HIP-event duration per launch, vs the gfx950 fp32 matrix peak.
            The generic kernel computes f32 products on the bf16 matrix pipe (three-way bf16 cut of
            both operands, six products, f32 accumulate: f32-equivalent), so the
            fraction is also given against that pipe's ceiling (2.5 PFLOP/s / 6).  MI_CONV_ARITH=f32
            selects the f32 MFMA instruction; its step time is reported next to the default's.
secondary = the inference half of the metric (voxels/s of sigmoid+NMS+top-K decode and of the DoG
            particle picker) with its own HBM roofline.
cpu_baseline = the CPU oracle (oracle/train_ref.py, torch fp32 on the host cores) on the same step.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

PEAK_F32_MATRIX_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD
PEAK_BF16_MATRIX_TFLOPS = 2500.0 # MI355X_MICROARCH.md: dense bf16 MFMA; one f32 product = 6 bf16 products
PEAK_HBM_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E spec peak
FLOP_PER_SUBTOMO = 3.66e9        # q fwd+bwd 2.70 + k fwd 0.96 GFLOP


def build_views(tomo_dev, n_crops, crop, seed):
    """Two views per crop centre, cut and z-normalised on the GPU by mi_crop_normalize: the crop, and
    the crop shifted by <=1 voxel and mirrored along x.
    Returns (pool_q, pool_k) of shape (n,1,c,c,c), resident in HBM."""
    from cet_pick_amd.datasets import subvols as S
    g = np.random.default_rng(seed)
    Z, H, W = tomo_dev.shape
    h = crop // 2
    centres = np.stack([g.integers(h + 1, W - h - 1, n_crops), g.integers(h + 1, H - h - 1, n_crops),
                        g.integers(h + 1, Z - h - 1, n_crops)], 1).astype(np.int32)
    shift = g.integers(-1, 2, (n_crops, 3)).astype(np.int32)
    pq = S.crop_znorm(tomo_dev, centres, (crop,) * 3)
    pk = S.crop_znorm(tomo_dev, centres + shift, (crop,) * 3, flip_x=True)
    return pq, pk


def conv_profile(engine, pq, pk, batch, steps):
    """Eager steps; every conv call is followed by PROFILE_REPEAT more launches of itself between two HIP
    events on the launch stream (back to back, so the eager-mode gap in front of a lone launch is not counted)."""
    from cet_pick_amd import hipops as H
    H.PROFILE = []
    saved = engine.use_graph
    engine.use_graph = False
    saved_overlap = engine.moco.overlap_key_branch
    engine.moco.overlap_key_branch = False      # launches timed one at a time, not sharing the chip with the key branch
    for i in range(steps):
        o = (i * batch) % (pq.shape[0] - batch + 1)
        engine.step(pq[o:o + batch], pk[o:o + batch])
    torch.cuda.synchronize()
    recs = H.PROFILE
    H.PROFILE = None
    engine.use_graph = saved
    engine.moco.overlap_key_branch = saved_overlap
    tot_ms = sum(e0.elapsed_time(e1) / r for _, _, e0, e1, r in recs)
    tot_flop = sum(f for _, f, _, _, _ in recs)
    by = {}
    for tag, f, e0, e1, r in recs:
        d = by.setdefault(tag, [0, 0.0, 0.0])
        d[0] += 1; d[1] += f; d[2] += e0.elapsed_time(e1) / r
    return tot_flop, tot_ms, len(recs), by


def cpu_baseline(batch, steps, seed):
    from oracle import train_ref as T
    from cet_pick_amd.synthetic import seeded_state_dict
    from cet_pick_amd.models.networks.moco_encoder_3d import TomoResClassifier3D, BasicBlock
    try:
        ncpu = len(os.sched_getaffinity(0))      # the cores this process may actually use
    except AttributeError:
        ncpu = os.cpu_count() or 1
    torch.set_num_threads(max(1, min(ncpu, 16)))   # the GPU box shares its host: 16 cores per GPU
    enc = TomoResClassifier3D(BasicBlock, [2, 2, 2, 2], {"proj": 256, "pred": 256}, 0)
    sd = {k: v.detach().clone().contiguous() for k, v in seeded_state_dict(enc, seed=seed).items()}
    for k in list(sd):
        if k.startswith("pred."):
            sd["proj." + k[5:]] = sd[k]
    g = torch.Generator().manual_seed(seed)
    queue = torch.nn.functional.normalize(torch.randn(128, 1024, generator=g), dim=0)
    ref = T.MocoRef(sd, queue, m=0.999, T=0.1, lr=1e-3)
    xq = torch.randn(batch, 1, 32, 32, 32, generator=g)
    xk = xq.flip(4)
    ref.step(xq, xk)                    # warm-up
    t0 = time.perf_counter()
    for _ in range(steps):
        ref.step(xq, xk)
    dt = time.perf_counter() - t0
    return batch * steps / dt, torch.get_num_threads()


def entry_point_record(batch, min_iters=50):
    """What `moco_main` delivers through its own loader: the C2 tomogram written as an MRC file + the
    reference's image list, `cet_pick_amd.moco_main.build(opt)` (encoders, MoCo, SGD, trainer = step engine,
    TomoFileMocoLoader: device load_rec / preprocess -> DoG picks -> crop centres), one warm-up epoch (it captures the
    hipGraph), then `trainer.train(epoch, loader)` - the real run_epoch with its meters - timed by the wall clock over
    >= `min_iters` iterations.  Sub-tomograms/s = iterations x batch / seconds.  N = 1 only."""
    import shutil
    import tempfile
    from cet_pick_amd import moco_main
    from cet_pick_amd.opts import opts
    from cet_pick_amd.synthetic import make_tomo
    from cet_pick_amd.utils import mrc
    cwd = os.getcwd()
    tmp = tempfile.mkdtemp(prefix="cetpick_entry_")
    try:
        os.chdir(tmp)
        os.makedirs("data")
        vol, _ = make_tomo((128, 512, 512), seed=317)
        mrc.write(os.path.join("data", "c2.rec"), vol)
        with open(os.path.join("data", "train_images.txt"), "w") as f:
            f.write("image_name\trec_path\nc2\tc2.rec\n")
        opt = opts().parse(["moco", "--arch", "moco3d_18", "--dataset", "simsiam3d", "--order", "zxy", "--batch_size", str(batch),
                            "--lr", "0.001", "--exp_id", "bench_entry", "--debug", "0", "--dog", "3,5", "--num_epochs", "1"])
        t_build = time.perf_counter()
        opt, model, optimizer, trainer, loader, _, _, _ = moco_main.build(opt)
        torch.cuda.synchronize()
        t_build = time.perf_counter() - t_build
        iters = len(loader)
        epochs = max(1, -(-min_iters // max(iters, 1)))
        loader.set_epoch(0)
        trainer.train(0, loader)                            # warm-up: eager steps, graph capture, allocator
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for e in range(1, epochs + 1):
            loader.set_epoch(e)
            log_dict, _ = trainer.train(e, loader)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        n_it = epochs * iters
        rec = {"value": n_it * batch / dt, "unit": "subtomograms/sec", "iterations": n_it, "epochs": epochs,
               "ms_per_iteration": dt / n_it * 1e3, "crop_centres": int(len(loader.centres)), "final_loss": float(log_dict["loss"]),
               "setup_s": t_build,
               "path": "python -m cet_pick_amd.moco_main moco --arch moco3d_18 --batch_size %d: MRC file -> device loader -> DoG "
                       "picks -> TomoFileMocoLoader (mi_crop_normalize_table, two launches per batch) -> BaseTrainer.run_epoch -> "
                       "MocoStepEngine (hipGraph)" % batch}
        trainer.close()
        return rec
    finally:
        os.chdir(cwd)
        shutil.rmtree(tmp, ignore_errors=True)


def _latest_profile(suffix):
    """profiles/rNN_<suffix> of the highest round present (the profile refresh of a round writes its own tag)"""
    import glob
    c = sorted(glob.glob(os.path.join(REPO, "profiles", "r[0-9][0-9]_" + suffix)))
    return c[-1] if c else None


def _traffic(fname, kernels):
    """HBM bytes per launch of `kernels` from a PMC summary under profiles/ (tools/pmc_summary.py), or None when the
    summary was measured on other kernel sources than the ones in the tree (its source hash no longer matches)."""
    path = os.path.join(REPO, "profiles", fname)
    if not os.path.exists(path):
        return None, "no %s" % fname
    from cet_pick_amd.build import source_sha16
    d = json.load(open(path))
    if d.get("source_sha16") != source_sha16(d.get("source_prefixes")):
        return None, "stale: %s was measured on sources %s" % (fname, d.get("source_sha16"))
    tot = 0.0
    if isinstance(kernels, tuple):
        # ("not", prefix, ...): EVERY kernel of the profiled run whose name starts with none of the prefixes, each with the
        # launches per call the summary recorded - the whole chain, fills and tail included
        for name, v in d["kernels"].items():
            if not name.startswith(kernels[1:]) and "hbm_bytes_per_launch" in v:
                tot += v["hbm_bytes_per_launch"] * v.get("launches_per_step", 1.0)
        return tot, "PMC FETCH_SIZE (x2, gfx950) + WRITE_SIZE, every launch of the chain, profiles/%s" % fname
    for k in kernels:
        hit = [v for name, v in d["kernels"].items() if name.startswith(k) and "hbm_bytes_per_launch" in v]
        if not hit:
            return None, "%s has no %s" % (fname, k)
        tot += sum(v["hbm_bytes_per_launch"] * kernels[k] for v in hit)
    return tot, "PMC FETCH_SIZE (x2, gfx950) + WRITE_SIZE per launch, profiles/%s" % fname


def inference_secondary(dev, with_cpu=True, rank=0, world=1):
    """voxels/s of the inference half on BASELINE configs[2] sizes, after the timed region.
    Each entry carries its own HBM roofline (8 B per voxel algorithmic) and CPU baseline(s).
    world > 1 (inference does not exchange - one tomogram per GPU): EVERY rank runs the chains on a
    tomogram of its own (seed 317 + rank) at the same time - the timed replays sit between two barriers - and the entry
    holds the SUM of the ranks' voxels/s, the slowest rank's ms and the aggregate rate over the wall clock of the
    concurrent region (which would show host-side launch or PCIe serialisation between the ranks)."""
    from cet_pick_amd.synthetic import make_tomo, make_logits
    from cet_pick_amd.models import decode as Dm
    from cet_pick_amd.utils import image as Im
    import torch.distributed as dist

    def barrier():
        if world > 1:
            torch.cuda.synchronize()
            dist.barrier()

    def timeit(fn, n, warm=3):
        """(eager ms, hipGraph-replay ms): the chain is a handful of short launches, so the product call is also timed
        captured in a hipGraph and replayed (launch-bound inner loops belong in graphs)."""
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        eager = e0.elapsed_time(e1) / n
        # captured: GREPS calls per graph - a replay has a fixed cost of its own (10 - 16 us on this part,
        # MI355X_MICROARCH.md "graph-replay-floor"), which a 50 us chain replayed one call at a time pays in full (round 2's
        # "ms_hipgraph > ms_eager" for the decode: eager launches are queued ahead of the GPU, single replays are not)
        GREPS = 8
        g = torch.cuda.CUDAGraph()
        # N > 1: the process group's watchdog thread polls its finished works with hipEventQuery at its own pace; inside a
        # "global" mode capture that call from another thread is fatal (MocoStepEngine._capture): thread_local mode cures
        # that.  The other race of a data-parallel capture (hipErrorCapturedEvent, MocoStepEngine._drain_watchdog) needs a
        # captured collective - RCCL's stream joining the capture - and this capture holds none, so no drain is needed HERE
        import torch.distributed as _dist
        pg = _dist.is_available() and _dist.is_initialized()
        with torch.cuda.graph(g, capture_error_mode="thread_local" if pg else "global"):
            for _ in range(GREPS):
                fn()
        g.replay()
        barrier()                              # N > 1: every rank replays its chain in the same window
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e0.record()
        for _ in range(max(1, n // GREPS)):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        barrier()
        calls = max(1, n // GREPS) * GREPS
        wall = (time.perf_counter() - t0) * 1e3 / calls
        return eager, e0.elapsed_time(e1) / calls, wall

    def entry(workload, n_vox, timing, kernels, traffic_file):
        eager, graph, wall = timing
        ms = min(eager, graph)
        rate, slow_ms, wall_ms = n_vox / ms * 1e3, ms, wall
        if world > 1:
            t = torch.tensor([rate, 0.0, 0.0], dtype=torch.float64, device=dev)
            dist.all_reduce(t)                                   # sum of the ranks' voxels/s
            m = torch.tensor([ms, wall], dtype=torch.float64, device=dev)
            dist.all_reduce(m, op=dist.ReduceOp.MAX)             # slowest rank; longest concurrent window
            rate, slow_ms, wall_ms = float(t[0]), float(m[0]), float(m[1])
        achieved = rate / world * 8 / 1e9                        # per GPU, against one GPU's HBM peak
        traffic, tnote = _traffic(traffic_file, kernels)
        e = {"workload": workload, "ms": round(slow_ms, 4), "ms_source": "eager" if eager <= graph else "hipgraph",
             "ms_eager": round(eager, 4), "ms_hipgraph": round(graph, 4),
             "voxels_per_sec": rate, "ranks": world,
             "roofline": {"bound": "hbm", "achieved": achieved, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                          "frac": achieved / PEAK_HBM_GBS, "traffic": traffic, "traffic_note": tnote,
                          "algorithmic_bytes": n_vox * 8,
                          "measured": "whole chain (every launch of the call) between two HIP events on the launch "
                                      "stream, per call; algorithmic 8 B per voxel (SURVEY.md 8d)"}}
        if world > 1:
            e["voxels_per_sec_wall"] = world * n_vox / wall_ms * 1e3
            e["note"] = ("one tomogram per rank (seed 317 + rank), the ranks' replays between two barriers: voxels_per_sec = sum "
                         "of the ranks' rates, ms = slowest rank, voxels_per_sec_wall = all ranks' voxels over the wall clock of "
                         "the concurrent window (barrier skew included); roofline per GPU")
        return e

    logits_np = make_logits((128, 256, 256), seed=317 + rank)
    logits = torch.as_tensor(logits_np).to(dev)[None, None]
    t_dec = timeit(lambda: Dm.sigmoid_tomo_decode(logits, kernel=3, K=900), 30)
    vol, _ = make_tomo((256, 512, 512), seed=317 + rank)
    v = torch.as_tensor(vol).to(dev)
    t_dog = timeit(lambda: Im.dog_pick(v, [3, 5]), 8)
    out = {
        "metric": "voxels/sec (heatmap+NMS)",
        "decode_sigmoid_nms_topk": entry("logits 1x1x128x256x256, k=3, K=900: sigmoid+clamp -> (3,3,3) NMS -> top-K", logits.numel(),
                                         t_dec, {"decode1_kernel": 1},
                                         INFER_TRAFFIC),
        "dog_pick": entry("tomogram 256x512x512, sigma=(3,5), nms_xy k=3, greedy d=14", v.numel(), t_dog,
                          DOG_KERNELS, INFER_TRAFFIC),
    }
    # the picker's second roofline: its filter stage is vector arithmetic - 198 multiply-adds per voxel over the
    # three passes of both Gaussians ((R1 + R2 + 2) per pass, R = 12 and 20) - against the f32 vector peak of MI355X_MICROARCH.md
    dflop = 198.0 * 2 * v.numel()
    dms = out["dog_pick"]["ms"]
    out["dog_pick"]["roofline_valu"] = {"bound": "valu", "achieved": dflop / dms / 1e9, "peak": PEAK_F32_MATRIX_TFLOPS, "unit": "TFLOP/s",
                                        "frac": dflop / dms / 1e9 / PEAK_F32_MATRIX_TFLOPS, "algorithmic_gflop": dflop / 1e9,
                                        "note": "whole chain time against the filter stage's 198 FMA per voxel (26.6 GFLOP at 256x512x512); the "
                                                "f32 vector peak equals the f32 matrix peak figure of the guide (157.3 TFLOP/s)"}
    if with_cpu:
        # CPU baselines on a bounded sample: the oracle (numpy / C port), and for the picker also the reference's own
        # arithmetic for the Gaussians - scipy.ndimage.gaussian_filter, single-threaded - in front of the oracle's tail.
        # Round 4: each also on ALL host cores - z-slabs with the halo the window needs in a thread pool
        # (numpy / scipy release the GIL in their loops); what is left serial (top-K, the greedy loop) is inside the timing.
        from concurrent.futures import ThreadPoolExecutor
        from oracle import infer_ref as O
        try:
            n_cores = len(os.sched_getaffinity(0))                 # the cores this process may actually use
        except AttributeError:
            n_cores = os.cpu_count() or 1
        n_cores = max(1, min(n_cores, 16))                         # (the GPU box shares its host: 16 cores per GPU)

        def slabs(fn, a, halo):
            """fn on z-slabs of `a` (+ halo planes where the volume has them), in threads; the slabs' interiors, concatenated."""
            d = a.shape[0]
            step = max(1, -(-d // n_cores))
            def one(z0):
                lo, hi = max(0, z0 - halo), min(d, z0 + step + halo)
                r = fn(a[lo:hi])
                return r[z0 - lo:z0 - lo + min(step, d - z0)]
            with ThreadPoolExecutor(n_cores) as ex:
                return np.concatenate(list(ex.map(one, range(0, d, step))), 0)
        t0 = time.perf_counter()
        hm = O.sigmoid_clamp(logits_np)
        O.tomo_decode(hm, kernel=3, K=900)
        dt = time.perf_counter() - t0
        out["decode_sigmoid_nms_topk"]["cpu_baseline"] = {
            "value": logits_np.size / dt, "unit": "voxels/sec", "cores": 1, "kind": "port",
            "sample": "the whole 128x256x256 logit volume once: oracle/infer_ref.py sigmoid_clamp + tomo_decode (numpy)"}
        t0 = time.perf_counter()
        hm = slabs(O.sigmoid_clamp, logits_np, 0)
        nmsd = slabs(lambda a: O.nms_window(a, (3, 3, 3)), hm, 1)
        sc, zz, yy, xx, _ = O.topk(nmsd, 900)
        dt = time.perf_counter() - t0
        out["decode_sigmoid_nms_topk"]["cpu_baseline_all_cores"] = {
            "value": logits_np.size / dt, "unit": "voxels/sec", "cores": n_cores, "kind": "port",
            "sample": "the same volume: sigmoid and the (3,3,3) NMS of the oracle on %d z-slabs in threads, top-K serial" % n_cores}
        sub = np.ascontiguousarray(vol[:64]).astype(np.float64)
        t0 = time.perf_counter()
        O.get_potential_coords_pyramid(sub, sigmas=(3, 5))
        dt = time.perf_counter() - t0
        out["dog_pick"]["cpu_baseline"] = {
            "value": sub.size / dt, "unit": "voxels/sec", "cores": 1, "kind": "port",
            "sample": "the first 64 slices (64x512x512) once: oracle/infer_ref.py get_potential_coords_pyramid (numpy fp64 + C greedy loop)"}
        try:
            from scipy import ndimage
            t0 = time.perf_counter()
            diff = ndimage.gaussian_filter(sub, 5) - ndimage.gaussian_filter(sub, 3)
            diff[:10] = 0; diff[-10:] = 0
            diff[:, :30, :] = 0; diff[:, -30:, :] = 0; diff[:, :, :30] = 0; diff[:, :, -30:] = 0
            heat = O.nms_window(diff, (1, 3, 3))
            O.non_maximum_suppression_3d(heat, 14, threshold=O.pos_threshold(heat))
            dt = time.perf_counter() - t0
            out["dog_pick"]["cpu_baseline_scipy"] = {
                "value": sub.size / dt, "unit": "voxels/sec", "cores": 1, "kind": "port",
                "sample": "same 64x512x512 sample with scipy.ndimage.gaussian_filter (the reference's Gaussian, single-threaded) "
                          "in front of the oracle's NMS / threshold / greedy tail"}
            # ... and on all cores: the Gaussians on z-slabs with a halo of 4 sigma + 1 planes (exact: the filter is truncated
            # at 4 sigma; the volume's own ends keep scipy's reflection), the NMS on slabs, the greedy loop serial
            full = vol.astype(np.float64)                                   # the whole 256x512x512 tomogram
            t0 = time.perf_counter()
            g5 = slabs(lambda a: ndimage.gaussian_filter(a, 5), full, 21)
            g3 = slabs(lambda a: ndimage.gaussian_filter(a, 3), full, 13)
            diff = g5 - g3
            diff[:10] = 0; diff[-10:] = 0
            diff[:, :30, :] = 0; diff[:, -30:, :] = 0; diff[:, :, :30] = 0; diff[:, :, -30:] = 0
            heat = slabs(lambda a: O.nms_window(a, (1, 3, 3)), diff, 0)
            O.non_maximum_suppression_3d(heat, 14, threshold=O.pos_threshold(heat))
            dt = time.perf_counter() - t0
            out["dog_pick"]["cpu_baseline_all_cores"] = {
                "value": full.size / dt, "unit": "voxels/sec", "cores": n_cores, "kind": "port",
                "sample": "the whole 256x512x512 tomogram once: scipy.ndimage.gaussian_filter and the oracle's xy-NMS on %d "
                          "z-slabs in threads, threshold + greedy loop (C) serial" % n_cores}
        except ImportError:
            pass
    return out


# PMC traffic summary of the inference chains (tools/pmc_run.sh + tools/pmc_summary.py) and the kernels of the picker's
# chain that it is summed over - EVERY launch of the call, not only the Gaussians
INFER_TRAFFIC = os.path.basename(_latest_profile("infer_traffic.json") or "r05_infer_traffic.json")
DOG_KERNELS = ("not", "decode1_", "peak3_", "topk_", "zero_header", "nms_march")      # = everything the picker launches

T_START = time.perf_counter()


def log(msg):
    print("[bench %7.1fs] %s" % (time.perf_counter() - T_START, msg), file=sys.stderr, flush=True)


def launch_ranks(args):
    """`python bench.py --gpus N` outside torch.distributed.run: this parent makes no GPU call (importing torch does not
    initialise HIP; torch.cuda.device_count() does not either on this image), starts N fresh ranks as a CHILD process
    (never an exec), relays rank 0's JSON line and exits with the child's code."""
    import socket
    import subprocess
    backend = os.environ.get("CETPICK_DIST_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if backend == "nccl" and ndev < args.gpus:
        log("--gpus %d asked for, %d visible: RCCL wants one GPU per rank (CETPICK_DIST_BACKEND=gloo shares one GPU "
            "as a rehearsal)" % (args.gpus, ndev))
        return 2
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    env.setdefault("OMP_NUM_THREADS", "4")
    log("launching %d ranks: %s" % (args.gpus, " ".join(cmd[1:])))
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)
    line = None
    for ln in proc.stdout:
        if ln.startswith("{") and '"metric"' in ln:
            line = ln.strip()
        else:
            sys.stderr.write(ln)
    rc = proc.wait()
    if line is not None:
        print(line, flush=True)
    elif rc == 0:
        rc = 1
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--crops", type=int, default=2048)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--no-conv-profile", action="store_true", help="skip the per-launch roofline pass (clean rocprofv3 runs of the step)")
    ap.add_argument("--no-entry-point", action="store_true", help="skip the moco_main entry-point record (MRC file -> loader -> run_epoch)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))

    # The one JSON line is the only thing this process writes to its stdout: RCCL prints a version banner to fd 1 at
    # communicator setup (and any other native library might), so fd 1 is pointed at stderr and the line goes out through a
    # saved duplicate of the original stdout.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    def emit(obj):
        os.write(json_fd, (json.dumps(obj) + "\n").encode())

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch.distributed as dist
    # CETPICK_BENCH_REHEARSE_RCCL=1 (one-GPU box): the N>1 code path - SyncBN sums, key all-gather, bucketed gradient
    # all-reduce, the collectives captured into the hipGraph, ordered tear-down - on a 1-rank RCCL group
    rehearse = world == 1 and os.environ.get("CETPICK_BENCH_REHEARSE_RCCL", "") == "1"
    if rehearse:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29655")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", local_rank))
    dist_active = world > 1 or rehearse
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # backend "nccl" IS RCCL on ROCm.  CETPICK_DIST_BACKEND=gloo is the one-GPU rehearsal of the N>1 path (ranks
        # share the device: RCCL refuses two ranks on one GPU)
        backend = os.environ.get("CETPICK_DIST_BACKEND", "nccl")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            local_rank = local_rank % max(torch.cuda.device_count(), 1)
            dist.init_process_group(backend)
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    import __graft_entry__ as ge
    if rank == 0:
        ge.build()
    if world > 1:
        dist.barrier()
    from cet_pick_amd import hipops as H
    from cet_pick_amd.synthetic import make_tomo
    from cet_pick_amd.models.networks.moco_encoder_3d import get_moco_net_small_3d
    from cet_pick_amd.models.moco import MoCo
    from cet_pick_amd.trains.moco_engine import MocoStepEngine

    log("extension built/loaded")
    torch.manual_seed(317)
    heads = {"proj": 256, "pred": 256}
    enc_q = get_moco_net_small_3d(18, heads, 0)
    enc_k = get_moco_net_small_3d(18, heads, 0)
    moco = MoCo(enc_q, enc_k, dim=128, r=1024, m=0.999, T=0.1).to(dev)
    if rehearse:
        H.FORCE_COLLECTIVES = True
    if dist_active:
        H.convert_sync_batchnorm(moco)
    moco.train()
    engine = MocoStepEngine(moco, lr=1e-3, use_graph=not args.no_graph)
    if dist_active:
        engine.broadcast_state(0)                   # identical replicas (parameter arenas, buffers, queue)

    # synthetic tomogram for this rank (seed 317 + rank), crops resident in HBM
    vol, _ = make_tomo((128, 512, 512), seed=317 + rank)
    tomo = torch.as_tensor(vol).to(dev)
    pq, pk = build_views(tomo, args.crops, 32, seed=317 + rank)
    B = args.batch
    nb = pq.shape[0] // B
    log("model + %d crop pairs resident on %s" % (pq.shape[0], dev))

    def run(i):
        o = (i % nb) * B
        return engine.step(pq[o:o + B], pk[o:o + B])

    for i in range(max(0, 3 - args.warmup)):            # set-up, not warm-up: the engine captures its hipGraph on the
        run(i)                                          # third call, which must not fall into the timed region
    for i in range(args.warmup):
        run(i)
        if i < 3:
            torch.cuda.synchronize()
            log("warm-up step %d done" % i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = run(args.warmup + i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    final_loss = float(loss.item())
    ranks_seen = 1
    if dist_active:
        ones = torch.ones(1, dtype=torch.float32, device=dev)     # every rank adds one: the number of ranks that really
        dist.all_reduce(ones)                                     # took part in the collectives of the timed steps
        ranks_seen = int(ones.item())
    log("timed region: %d steps in %.3f s" % (args.steps, dt))

    out = None
    # the roofline pass runs three more steps: EVERY rank takes them (they contain the step's collectives)
    step_nodes = engine.node_counts() if engine.use_graph else None
    if args.no_conv_profile:
        if rank == 0:
            emit({"metric": "subtomograms/sec (MoCo-3D train) + voxels/sec (heatmap+NMS) at 1/2/4/8 GPU",
                  "value": B * world * args.steps / dt, "unit": "subtomograms/sec", "n_gpus": world,
                  "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
                  "step_graph_nodes": step_nodes, "note": "--no-conv-profile: no roofline pass"})
        if dist_active:
            dist.barrier()
            engine.close()
            dist.destroy_process_group()
        return
    flop, ms, n_launch, by = conv_profile(engine, pq, pk, B, 3)
    arith = "f32" if os.environ.get("MI_CONV_ARITH", "")[:1] == "f" else "bf16x3"
    # the same step on the f32 MFMA instruction (fresh graph capture; N = 1 only: a reported comparison, not `value`)
    f32_ms = None
    if not dist_active and arith == "bf16x3" and not args.no_secondary:
        os.environ["MI_CONV_ARITH"] = "f32"
        engine.close()
        for i in range(3):
            run(i)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(args.steps):
            run(i)
        torch.cuda.synchronize()
        f32_ms = (time.perf_counter() - t1) / args.steps * 1e3
        os.environ.pop("MI_CONV_ARITH")
        engine.close()
    if rank == 0:
        value = B * world * args.steps / dt
        achieved = flop / (ms * 1e-3) / 1e12
        peak = PEAK_BF16_MATRIX_TFLOPS / 6 if arith == "bf16x3" else PEAK_F32_MATRIX_TFLOPS
        out = {
            "metric": "subtomograms/sec (MoCo-3D train) + voxels/sec (heatmap+NMS) at 1/2/4/8 GPU",
            "value": value, "unit": "subtomograms/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "rccl_ranks": ranks_seen, "dist_backend": (dist.get_backend() if dist_active else None),
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "moco_main.py moco3d_18: 3D encoder, synthetic 512x512x128 tomogram, 32^3 subtomo "
                                   "crops, batch 64 per GPU, r=1024, dim=128, m=0.999, T=0.1, SGD lr 1e-3",
                       "inputs": "the crop pairs of the run are cut once and stay resident in HBM (no host-to-device copy "
                                 "inside the timed region; the CPU baseline is timed the same way)",
                       "global_batch": B * world, "parallelism": "dp%d" % world,
                       "conv_arithmetic": ("f32 products as 6 bf16 MFMA products of a 3-way bf16 cut, f32 accumulate "
                                           "(f32-equivalent; MI_CONV_ARITH=f32 for the f32 MFMA)" if arith == "bf16x3"
                                           else "v_mfma_f32_32x32x2_f32"),
                       "hipgraph": bool(engine.use_graph), "final_loss": final_loss},
            "step_graph_nodes": step_nodes,
            "step_mfma_frac_of_peak": value / world * FLOP_PER_SUBTOMO / 1e12 / peak,
            "roofline": {"bound": "mfma", "kernel": "the conv family: direct3 / direct3s / direct3_wgrad / cube2 / pair_wgrad / stem_fwd_bf3 / stem_wgrad_bf3 / conv_igemm (+ split-K reduces): all 75 conv calls of a step, forward / data gradient / weight gradient",
                         "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                         "frac": achieved / peak, "traffic": None,
                         "peak_note": ("peak = the pipe the kernel runs on: the bf16x3 arithmetic executes 6 bf16 MFMA "
                                       "products per f32 product, so f32-equivalent work tops out at 2500/6 TFLOP/s; "
                                       "frac_f32_mfma_peak is the same rate against the f32 MFMA peak (157.3)"
                                       if arith == "bf16x3" else "peak = f32 MFMA (v_mfma_f32_32x32x2_f32)"),
                         "frac_f32_mfma_peak": achieved / PEAK_F32_MATRIX_TFLOPS,
                         # tools/probes/mfma_bf16_peak.hip, round 4: v_mfma_f32_32x32x16_bf16 from registers on the whole chip sustains
                         # 2,130 TFLOP/s (2.05 GHz under MFMA load) - what the bf16x3 arithmetic can reach on this part; `frac` stays
                         # against the nominal figure
                         "frac_of_sustained_mfma": (achieved / (2130.0 / 6.0)) if arith == "bf16x3" else None,
                         "launches_per_step": n_launch // 3, "kernel_ms_per_step": ms / 3,
                         "algorithmic_gflop_per_step": flop / 3 / 1e9,
                         "by_mode": {t: {"launches_per_step": v[0] // 3, "gflop_per_step": v[1] / 3 / 1e9,
                                         "ms_per_step": v[2] / 3, "tflops": v[1] / (v[2] * 1e-3) / 1e12}
                                     for t, v in sorted(by.items())},
                         "measured": "per conv call of 3 eager steps after the timed region: 8 back-to-back launches of the "
                                     "call (kernel + its split-K reduce) between two HIP events on the launch stream; "
                                     "compare profiles/%s" % os.path.basename(_latest_profile("train_kernel_stats.csv") or "r05_train_kernel_stats.csv")},
        }
        # HBM traffic of the conv kernels from PMC counters (tools/pmc_run.sh + tools/pmc_summary.py: separate rocprofv3
        # --pmc passes of this workload; FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md), per conv call.
        # The summary carries the hash of the kernel sources it was measured on: a stale one is not quoted.
        tpath = _latest_profile("conv_traffic.json")
        if tpath:
            from cet_pick_amd.build import source_sha16
            tj = json.load(open(tpath))
            if tj.get("source_sha16") == source_sha16(tj.get("source_prefixes")):
                # bytes PER STEP (a per-call average over 61 heterogeneous launches is a unit nobody can use);
                # SURVEY.md 8(d) estimates ~1.6 GB per step (EMA + SGD + weights + activations)
                out["roofline"]["traffic"] = tj["hbm_bytes_per_step"]
                out["roofline"]["traffic_per_conv_call"] = tj["hbm_bytes_per_step"] / (n_launch // 3)
                out["roofline"]["traffic_ratio_to_estimate"] = tj["hbm_bytes_per_step"] / 1.6e9
                out["roofline"]["traffic_note"] = ("HBM bytes per training step of the conv family (%d calls), PMC FETCH_SIZE (x2, gfx950) + "
                                                   "WRITE_SIZE, profiles/%s; ratio to SURVEY 8(d)'s 1.6 GB estimate beside it: the step moves "
                                                   "~1.4 TB/s, far from the HBM roof - MFMA stays the binding bound"
                                                   % (n_launch // 3, os.path.basename(tpath)))
            else:
                out["roofline"]["traffic_note"] = "stale: profiles/%s was measured on other kernel sources" % os.path.basename(tpath)
        if f32_ms is not None:
            out["f32_mfma_step"] = {"ms_per_step": f32_ms, "value": B / (f32_ms * 1e-3),
                                    "note": "the same step with MI_CONV_ARITH=f32 (v_mfma_f32_32x32x2_f32 in the generic kernel)"}
        log("conv roofline pass done")
    # the voxels/s half of the metric: N = 1 on the one rank; N > 1 on EVERY rank at once (one tomogram per GPU, no exchange)
    if not args.no_secondary and (world > 1 or rank == 0):
        sec = inference_secondary(dev, with_cpu=(not args.no_cpu_baseline) and world == 1, rank=rank, world=world)
        if rank == 0:
            out["secondary"] = sec
            log("inference secondary done")
    if rank == 0:
        if not args.no_secondary and world == 1:
            # detector-side rows (loader a12, unet_4 forward a22, debiased contrastive loss a23, C5 train step)
            from tools.bench_detector import run as detector_secondary
            out["secondary"]["detector"] = detector_secondary()
            log("detector secondary done")
            # the other half of the train path north_star names: the SimSiam-2D loop at the reference's documented exploration
            # configuration (docs/explore.md:67: simsiam2d_18, --bbox 36, batch 256, SGD lr 1e-3), tools/bench_simsiam2d.py
            from tools.bench_simsiam2d import run as simsiam2d_record
            out["secondary"]["simsiam2d_train_step"] = simsiam2d_record(with_cpu=not args.no_cpu_baseline)
            log("SimSiam-2D secondary done")
        if not args.no_cpu_baseline and world == 1:
            v, cores = cpu_baseline(B, 4, 317)
            out["cpu_baseline"] = {"value": v, "unit": "subtomograms/sec", "cores": cores, "kind": "port",
                                   "sample": "4 MoCo steps of batch 64 (after 1 warm-up) of the same workload "
                                             "with oracle/train_ref.py (torch fp32, all host cores)"}
    if rank == 0 and world == 1 and not dist_active and not args.no_entry_point:
        engine.close()                              # (its graph and buffers are no longer needed: the record builds its own)
        out["entry_point"] = entry_point_record(B)
        out["entry_point"]["ratio_to_value"] = out["entry_point"]["value"] / out["value"]
        log("entry-point record done")
        # ... and what the inference entry points deliver on the C3 volume read from an MRC file (tools/bench_infer_entry.py)
        from tools.bench_infer_entry import run as infer_entry_record
        out["entry_point_infer"] = infer_entry_record()
        log("inference entry-point record done")
    if rank == 0:
        # the whole metric as top-level scalars, in front of the long nested blocks (the driver's parsed record keeps these)
        head = {}
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step"):
            head[k] = out[k]
        sec = out.get("secondary") or {}
        dec, dog = sec.get("decode_sigmoid_nms_topk"), sec.get("dog_pick")
        head["voxels_per_sec_decode"] = dec["voxels_per_sec"] if dec else None
        head["voxels_per_sec_dog"] = dog["voxels_per_sec"] if dog else None
        head["decode_ms"] = dec["ms"] if dec else None
        head["dog_ms"] = dog["ms"] if dog else None
        head["f32_mfma_ms_per_step"] = out["f32_mfma_step"]["ms_per_step"] if "f32_mfma_step" in out else None
        head["entry_point_value"] = out["entry_point"]["value"] if "entry_point" in out else None
        head["unet4_forward_ms"] = (sec.get("detector") or {}).get("unet4_forward", {}).get("ms")
        head["semi_train_step_ms"] = (sec.get("detector") or {}).get("semi_train_step", {}).get("ms")
        head["entry_point_infer_tot_ms"] = (out.get("entry_point_infer") or {}).get("test_py_detector", {}).get("tot_ms")
        head["simsiam2d_train_step_ms"] = (sec.get("simsiam2d_train_step") or {}).get("ms")
        head["simsiam2d_crop_pairs_per_sec"] = (sec.get("simsiam2d_train_step") or {}).get("crop_pairs_per_sec")
        head.update({k: v for k, v in out.items() if k not in head})
        out = head
        emit(out)                                    # the line is out before any tear-down
    if dist_active:
        # tear-down in dependency order: the captured hipGraph (it holds the RCCL kernels and the events of the async
        # work handles recorded during capture) goes before the communicator it refers to
        rc = 0
        try:
            dist.barrier()
            engine.close()
            dist.destroy_process_group()
        except Exception:
            import traceback
            traceback.print_exc()
            rc = 3                                   # a failed tear-down is a failed run, never a silent exit 0
        sys.stdout.flush()
        sys.stderr.flush()
        if rc:
            sys.exit(rc)


if __name__ == "__main__":
    main()
