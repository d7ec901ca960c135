"""Per-layer timing of conv_igemm over tile/slice/split overrides (MI_CONV_BM / MI_CONV_BK / MI_CONV_SPLITS)."""
import os, sys, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cet_pick_amd import hipops as H

LAYERS = {
    "stem": (64, 32, 32, 32, 1, 64, 7, 2, 3),
    "l1":   (64, 8, 8, 8, 64, 64, 3, 1, 1),
    "l2a":  (64, 8, 8, 8, 64, 128, 3, 2, 1),
    "l2b":  (64, 4, 4, 4, 128, 128, 3, 1, 1),
    "l3a":  (64, 4, 4, 4, 128, 256, 3, 2, 1),
    "l3b":  (64, 2, 2, 2, 256, 256, 3, 1, 1),
    "ds2":  (64, 8, 8, 8, 64, 128, 1, 2, 0),
}
def kernels_table(crop, batch):
    """Which kernel family every convolution of the MoCo-3D encoder takes at a crop size, forward / data gradient / weight
    gradient, with its time (us per call, hipGraph replay of 4 calls):  python tools/bench_conv.py --kernels [crop] [batch]"""
    from cet_pick_amd import _lib as L
    lib = L.lib()
    def timeit(fn, n=20):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(4): fn()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): g.replay()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n / 4 * 1e3
    s1 = crop // 2
    g1, g2, g3 = crop // 4, crop // 8, crop // 16
    layers = [("conv1 (stem)", (s1 * 2,) * 3, 1, 64, 7, 2, 3, None),
              ("layer1 x4", (g1,) * 3, 64, 64, 3, 1, 1, None),
              ("layer2.0 front", (g1,) * 3, 64, 128, 3, 2, 1, "ds"),
              ("layer2 x3", (g2,) * 3, 128, 128, 3, 1, 1, None),
              ("layer3.0 front", (g2,) * 3, 128, 256, 3, 2, 1, "ds"),
              ("layer3 x3 + feature_3d", (g3,) * 3, 256, 256, 3, 1, 1, None)]
    print("crop %d^3, batch %d" % (crop, batch))
    print("%-24s %-6s %-26s %9s" % ("layer", "pass", "kernel", "us"))
    for name, (d, h, w), ci, co, k, st, pd, ds in layers:
        x = torch.randn(batch, d, h, w, ci, device="cuda")
        wt = H.conv_weight_param(co, ci, k); wt.data = wt.data.cuda(); wt.data.normal_()
        wd = None
        if ds:
            wd = H.conv_weight_param(co, ci, 1); wd.data = wd.data.cuda(); wd.data.normal_()
        y = H.conv_fwd(x, wt, k, st, pd)
        dy = torch.randn_like(y)
        def last():
            return lib.mi_debug_last_conv_kernel().decode()
        rows = []
        if ds and H.conv_fwd_s2_block(x, wt, wd) is not None:
            rows.append(("fwd", "s2_fwd (conv + shortcut)", timeit(lambda: H.conv_fwd_s2_block(x, wt, wd))))
        else:
            H.conv_fwd(x, wt, k, st, pd); kn = last()
            rows.append(("fwd", kn, timeit(lambda: H.conv_fwd(x, wt, k, st, pd))))
            if ds:
                H.conv_fwd(x, wd, 1, st, 0); kn = last()
                rows.append(("fwd", "shortcut: " + kn, timeit(lambda: H.conv_fwd(x, wd, 1, st, 0))))
        if ci > 1:
            if ds and H.conv_dgrad_s2_block(dy, dy, wt, wd, x.shape) is not None:
                rows.append(("dgrad", "s2_dgrad (conv + shortcut)", timeit(lambda: H.conv_dgrad_s2_block(dy, dy, wt, wd, x.shape))))
            else:
                H.conv_dgrad(dy, wt, x.shape, k, st, pd); kn = last()
                rows.append(("dgrad", kn, timeit(lambda: H.conv_dgrad(dy, wt, x.shape, k, st, pd))))
        def wg():
            wt.grad = None
            H.conv_wgrad_into(x, dy, wt, k, st, pd)
        wg(); kn = last()
        rows.append(("wgrad", kn, timeit(wg)))
        for ps, kn, us in rows:
            print("%-24s %-6s %-26s %9.1f" % (name, ps, kn, us), flush=True)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--kernels":
        kernels_table(int(sys.argv[2]) if len(sys.argv) > 2 else 32, int(sys.argv[3]) if len(sys.argv) > 3 else 64)
        return
    sel = sys.argv[1].split(",") if len(sys.argv) > 1 else list(LAYERS)
    modes = sys.argv[2].split(",") if len(sys.argv) > 2 else ["fwd", "dgrad", "wgrad"]
    bms = [int(v) for v in (sys.argv[3].split(",") if len(sys.argv) > 3 else ["64", "128"])]
    bks = [int(v) for v in (sys.argv[4].split(",") if len(sys.argv) > 4 else ["16", "32"])]
    spl = [int(v) for v in (sys.argv[5].split(",") if len(sys.argv) > 5 else ["0"])]
    bns = [int(v) for v in (sys.argv[6].split(",") if len(sys.argv) > 6 else ["64"])]


    def timeit(fn, n=20):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n

    for name in sel:
        n, d, h, w, ci, co, k, s, p = LAYERS[name]
        x = torch.randn(n, d, h, w, ci, device="cuda")
        wt = H.conv_weight_param(co, ci, k); wt.data = wt.data.cuda(); wt.data.normal_()
        y = H.conv_fwd(x, wt, k, s, p)
        dy = torch.randn_like(y)
        flop = 2.0 * y.numel() * ci * k ** 3
        for mode in modes:
            if mode == "dgrad" and ci == 1: continue
            for bm, bk, sp, bn in itertools.product(bms, bks, spl, bns):
                os.environ["MI_CONV_BM"] = str(bm); os.environ["MI_CONV_BK"] = str(bk); os.environ["MI_CONV_BN"] = str(bn)
                if sp: os.environ["MI_CONV_SPLITS"] = str(sp)
                else: os.environ.pop("MI_CONV_SPLITS", None)
                if mode == "fwd": fn = lambda: H.conv_fwd(x, wt, k, s, p)
                elif mode == "dgrad": fn = lambda: H.conv_dgrad(dy, wt, x.shape, k, s, p)
                else:
                    wt.grad = None
                    fn = lambda: (setattr(wt, "grad", None), H.conv_wgrad_into(x, dy, wt, k, s, p))
                ms = timeit(fn)
                print("%-5s %-5s bm=%3d bn=%3d bk=%2d splits=%2d  %8.1f us  %6.1f TF/s" % (name, mode, bm, bn, bk, sp, ms * 1e3, flop / ms / 1e9), flush=True)


if __name__ == "__main__":
    main()
