"""Reproducer for the process-group watchdog abort with captured collectives (1-rank RCCL group).
  drop: the Work of a collective captured into a hipGraph is dropped right away; its events go back to the process
        group's event cache and are recycled by the next EAGER collective, whose Work the watchdog thread then polls.
  keep: the captured Work is kept alive (what hipops.CAPTURED_WORKS does).
Each mode: capture, a few eager collectives + replays, 1 s for the watchdog to poll.  Exit code 0 = survived."""
import os, sys, time
import torch, torch.distributed as dist
mode = sys.argv[1] if len(sys.argv) > 1 else "drop"
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29671")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
x = torch.ones(1 << 16, device="cuda")
for _ in range(3):
    dist.all_reduce(x)
torch.cuda.synchronize()
time.sleep(0.5)
keep = []
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(8):
        w = dist.all_reduce(x, async_op=True)
        w.wait()
        if mode == "keep":
            keep.append(w)
        del w
    x.mul_(1.0)
torch.cuda.synchronize()
for i in range(20):
    y = torch.ones(4, device="cuda")
    dist.all_reduce(y)              # eager: may take a recycled event
    g.replay()
    g.replay()
torch.cuda.synchronize()
time.sleep(1.0)
print("survived", mode, flush=True)
g.reset()
keep.clear()
dist.destroy_process_group()
print("torn down", mode, flush=True)
