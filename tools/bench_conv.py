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
def main():
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
