"""Counts inside one mi_dog_pick call (candidate list sizes, picks, what the chip-wide passes left open) read back from the
workspace header (GreedyHeader, infer_greedy.hip), found by searching the workspace for the cutoff's bit pattern."""
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
ws = L.workspace(0, v.device, "dog")
words = ws[: ws.numel() // 4 * 4].view(torch.int32)
cutbits = int(np.float32(cut.item()).view(np.int32))
hits = (words == cutbits).nonzero().flatten().cpu().numpy()
for h in hits:
    hdr = words[h - 5:h + 19].cpu().numpy().view(np.uint32)
    if hdr[1] == 0 or hdr[2] != int(n.item()):
        continue
    print("header @word %d: cand_count=%d n=%d n_kept=%d n_left=%d overflow=%d n_runs=%d ticket=%d trace=%s"
          % (h - 5, hdr[0], hdr[1], hdr[2], hdr[3], hdr[4], hdr[6], hdr[7], list(hdr[8:24])))
print("picks", int(n.item()), "cutoff", float(cut.item()))
