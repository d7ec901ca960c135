"""Counts inside one mi_dog_pick call (candidate list sizes, picks) read back from the workspace header, and the cost
of a trivial launch (the floor every extra kernel of the chain pays).  Layout mirrors dog_ws_layout (infer_greedy.hip)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cet_pick_amd import _lib as L
from cet_pick_amd.synthetic import make_tomo
from cet_pick_amd.utils import image as Im

D, H, W = (256, 512, 512) if len(sys.argv) < 4 else tuple(int(v) for v in sys.argv[1:4])
vol, _ = make_tomo((D, H, W), seed=317)
v = torch.as_tensor(vol).cuda()
s, c, n, cut, _ = Im.dog_pick(v, [3, 5])
torch.cuda.synchronize()
al = lambda x: (x + 255) // 256 * 256
nv = D * H * W
chunk = H
while chunk > 32 and D * ((H + chunk - 1) // chunk) < 2048:
    chunk = (chunk + 1) // 2
n_seg = D * ((H + chunk - 1) // chunk)
seg_cap = chunk * 512 // 4 + 64
off = 4 * al(4 * nv) + al(8 * max(nv // 4 + 1024, n_seg * seg_cap))
if len(sys.argv) > 4:
    off = int(sys.argv[4])
ws = L.workspace(0, v.device, "dog")
# the stats block sits between the candidates and the header: search the header by its cutoff field
raw = ws[off:off + 1024 * 1024].cpu().numpy().view(np.uint32)
cutbits = np.float32(cut.item()).view(np.uint32)
hits = np.nonzero(raw == cutbits)[0]
for h in hits[:3]:
    hdr = raw[h - 5:h + 3]
    print("header @+%d: cand_count=%d n=%d n_kept=%d n_deltas=%d overflow=%d n_runs=%d" % (4 * (h - 5), hdr[0], hdr[1], hdr[2], hdr[3], hdr[4], hdr[6]))
print("picks", int(n.item()), "cutoff", float(cut.item()))

lib = L.lib()
buf = torch.empty(16384, dtype=torch.uint8, device="cuda")
st = L.stream()
for _ in range(10):
    lib.mi_decode_workspace_init(L.ptr(buf), buf.numel(), st)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200):
    lib.mi_decode_workspace_init(L.ptr(buf), buf.numel(), st)
e1.record()
torch.cuda.synchronize()
print("trivial kernel, 200 back-to-back eager launches: %.2f us each" % (e0.elapsed_time(e1) / 200 * 1e3))
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(50):
        lib.mi_decode_workspace_init(L.ptr(buf), buf.numel(), L.stream())
g.replay(); torch.cuda.synchronize()
e0.record()
for _ in range(4):
    g.replay()
e1.record()
torch.cuda.synchronize()
print("trivial kernel, 50 per hipGraph: %.2f us each" % (e0.elapsed_time(e1) / 200 * 1e3))
# round trace (GreedyHeader.trace): candidates still open at the start of each chip-wide round launch; [14] = left for the finisher
for h in hits[:1]:
    tr = raw[h + 3:h + 3 + 16].tolist()
    print("undecided at the start of each round launch:", tr[:8], "| lists over capacity", tr[13],
          "| left for the finisher", tr[14])
