"""Three unet_4 forwards on 128 x 512 x 512 (run under tools/gprof.sh for the kernel table of the detector's C3 configuration)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from cet_pick_amd.models.model import create_model
from cet_pick_amd.synthetic import seeded_state_dict

net = create_model("unet_4", {"hm": 1, "proj": 32}, 32)
net.load_state_dict(seeded_state_dict(net, seed=321))
net = net.cuda().eval()
vol = torch.randn(1, 128, 512, 512, device="cuda")
with torch.no_grad():
    for _ in range(3):
        net(vol)
torch.cuda.synchronize()
