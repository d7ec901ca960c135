"""Diagnostic: ucl backward at 2N = 196,608 against dense float64 rows, row side and column side apart."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cet_pick_amd.models.loss import _UclRowSumsFn
n2, dim, T = int(sys.argv[1]) if len(sys.argv) > 1 else 196608, 32, 0.07
g = torch.Generator().manual_seed(0)
f32 = torch.nn.functional.normalize(torch.randn(n2, dim, generator=g), dim=1).cuda()
cls = torch.randint(0, 4, (n2,), generator=g).to(torch.uint8).cuda()
fd = f32.double()
half = n2 // 2
rows = torch.arange(5, n2, max(n2 // 40, 1), device="cuda")
ar = torch.arange(rows.numel(), device="cuda")
pair = (rows + half) % n2
posd, othd = (cls & 1).double(), ((cls >> 1) & 1).double()
for tag, scales, active in (("g_all only, all rows", (1, 0, 0, 0), None), ("g_all only, rows < N active", (1, 0, 0, 0), "lo"),
                            ("g_pos only", (0, 1, 0, 0), None), ("g_pair only", (0, 0, 0, 1), None), ("all four", (1, 3, 2, 50), None)):
    gen = torch.Generator(device="cuda").manual_seed(11)
    ups = [torch.rand(n2, device="cuda", generator=gen) * sc for sc in scales]
    if active == "lo":
        for u in ups: u[half:] = 0
    fg = f32.clone().requires_grad_()
    m, a2, p2, o2, e2 = _UclRowSumsFn.apply(fg, cls, 1.0 / T)
    (a2 * ups[0] + p2 * ups[1] + o2 * ups[2] + e2 * ups[3]).sum().backward()
    got = fg.grad[rows].double()
    U = [u.double() for u in ups]
    md = m.double()
    S = (fd[rows] @ fd.t()) / T
    Er = torch.exp(S - md[rows][:, None])
    Wr = Er * (U[0][rows][:, None] + U[1][rows][:, None] * posd[None, :] + U[2][rows][:, None] * othd[None, :])
    Wr[ar, pair] += U[3][rows] * Er[ar, pair]
    Ec = torch.exp(S - md[None, :])
    Wc = Ec * (U[0][None, :] + U[1][None, :] * posd[rows][:, None] + U[2][None, :] * othd[rows][:, None])
    Wc[ar, pair] += U[3][pair] * Ec[ar, pair]
    Wr[ar, rows] = 0; Wc[ar, rows] = 0
    wr, wc = (Wr @ fd) / T, (Wc @ fd) / T
    want = wr + wc
    sc = want.abs().max(1, keepdim=True)[0]
    e = (got - want)
    # projection of the error on the row part and the column part (least squares per sampled row)
    A = torch.stack([wr, wc], 2)                                   # rows x dim x 2
    coef = torch.linalg.lstsq(A, e.unsqueeze(2)).solution.squeeze(2)
    print("%-28s max rel err %.2e | error ~ %.2e x row part + %.2e x column part (median coefficients); m - true max: %.2e" %
          (tag, float((e.abs() / sc).max()), float(coef[:, 0].median()), float(coef[:, 1].median()),
           float((md[rows] - S.max(1)[0]).abs().max())))
