"""Diagnostic for the process-group tear-down with a captured data-parallel step (1-rank RCCL group, every collective
forced on).  `python tools/diag_teardown.py {noclose|close}`: `noclose` destroys the group while the hipGraph that holds the
captured RCCL work is still alive (what round 1 hid behind os._exit), `close` releases the graph first (engine.close()).
Prints what happened; the exit code is the tear-down's."""
import os
import sys
import traceback

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mode = sys.argv[1] if len(sys.argv) > 1 else "close"
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29641")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from cet_pick_amd import hipops as H
from cet_pick_amd.models.moco import MoCo
from cet_pick_amd.models.networks.moco_encoder_3d import get_moco_net_small_3d
from cet_pick_amd.trains.moco_engine import MocoStepEngine

H.FORCE_COLLECTIVES = True
torch.manual_seed(5)
heads = {"proj": 256, "pred": 256}
moco = MoCo(get_moco_net_small_3d(18, heads, 0), get_moco_net_small_3d(18, heads, 0), dim=128, r=256).cuda()
H.convert_sync_batchnorm(moco)
moco.train()
engine = MocoStepEngine(moco, lr=1e-2, use_graph=True)
x = torch.randn(8, 1, 32, 32, 32, device="cuda")
for i in range(6):
    engine.step(x, x.flip(4))
torch.cuda.synchronize()
print("steps done, graph captured:", engine._graph is not None, flush=True)
rc = 0
try:
    if mode == "close":
        engine.close()
    dist.destroy_process_group()
    print("destroy_process_group returned (%s)" % mode, flush=True)
except BaseException:
    traceback.print_exc()
    rc = 3
sys.stdout.flush()
sys.exit(rc)
