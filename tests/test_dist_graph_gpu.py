"""The data-parallel step (SyncBN sums, key all-gather, bucketed gradient all-reduce) captured into a hipGraph together
with its RCCL collectives.  One GPU is all a test box has, so the collectives run on a 1-rank RCCL group
(hipops.FORCE_COLLECTIVES issues them although the world size is 1): the captured step must reproduce the eager
data-parallel step and the plain single-process step."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import json, os, sys
import torch, torch.distributed as dist
sys.path.insert(0, %(repo)r)
mode = sys.argv[1]
torch.cuda.set_device(0)
if mode != "single":
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from cet_pick_amd import hipops as H
from cet_pick_amd.models.networks.moco_encoder_3d import get_moco_net_small_3d
from cet_pick_amd.models.moco import MoCo
from cet_pick_amd.trains.moco_engine import MocoStepEngine
if mode != "single":
    H.FORCE_COLLECTIVES = True
torch.manual_seed(5)
heads = {"proj": 256, "pred": 256}
moco = MoCo(get_moco_net_small_3d(18, heads, 0), get_moco_net_small_3d(18, heads, 0), dim=128, r=256, m=0.99, T=0.1).cuda()
if mode != "single":
    H.convert_sync_batchnorm(moco)
moco.train()
engine = MocoStepEngine(moco, lr=1e-2, use_graph=(mode != "eager"))
g = torch.Generator(device="cuda").manual_seed(11)
xs = [torch.randn(8, 1, 32, 32, 32, device="cuda", generator=g) for _ in range(6)]
losses = []
for i in range(6):
    losses.append(float(engine.step(xs[i], xs[i].flip(4))))
torch.cuda.synchronize()
out = {"losses": losses, "graph": engine._graph is not None, "buckets": engine.buckets_sent,
       "w": float(engine.arena_q.flat.double().abs().sum()), "k": float(engine.arena_k.flat.double().abs().sum()),
       "queue": float(moco.queue.double().abs().sum())}
print("RESULT " + json.dumps(out), flush=True)
# tear-down in dependency order: the graph that holds the captured RCCL work goes before the communicator
engine.close()
if mode != "single":
    dist.destroy_process_group()
print("TEARDOWN ok", flush=True)
'''


def free_port():
    """A port nobody listens on right now (a fixed number can still be in TIME_WAIT from an earlier run on the same box)."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def run(mode, port=None):
    port = port or free_port()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    if mode == "eager":
        env["CETPICK_DIST_GRAPH"] = "0"
    r = subprocess.run([sys.executable, "-c", SCRIPT % {"repo": REPO}, mode], env=env, capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0, "stdout tail: " + r.stdout[-600:] + "\nstderr tail: " + r.stderr[-3000:]
    assert "TEARDOWN ok" in r.stdout, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1]
    import json
    return json.loads(line[7:])


def test_data_parallel_step_is_captured_with_its_collectives():
    graph = run("graph")
    eager = run("eager")
    single = run("single")
    assert graph["graph"], "the data-parallel step did not end up in a hipGraph"
    assert not eager["graph"] and single["graph"]
    assert graph["buckets"] == ["layer3", "layer2", "layer1", "stem"] == eager["buckets"]
    # captured == eager, bit for bit: same kernels, same order, the collectives of a 1-rank group are identities
    assert graph["losses"] == eager["losses"], (graph["losses"], eager["losses"])
    for k in ("w", "k", "queue"):
        assert graph[k] == eager[k], (k, graph[k], eager[k])
    # against the plain single-process step only the first loss is comparable: the SyncBN path computes its statistics in
    # a launch of its own (the single-process run uses the one-launch small-batch kernel), a last-ulp difference that
    # this learning rate amplifies from the second step on
    assert abs(graph["losses"][0] - single["losses"][0]) <= 1e-5 * abs(single["losses"][0])


def test_bench_n_gt_1_path_on_one_rank_rccl_group_tears_down():
    """bench.py's N>1 code path (captured collectives, rccl_ranks, ordered tear-down with destroy_process_group) on the
    one GPU of a test box: a 1-rank RCCL group with every collective forced on.  A failed tear-down is a non-zero exit."""
    import json
    env = dict(os.environ, CETPICK_BENCH_REHEARSE_RCCL="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--steps", "10", "--warmup", "4", "--no-secondary", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    # no retry: a failure here is reported with the child's output (profiles/r04_watchdog_abort.txt was found that way)
    assert r.returncode == 0, "exit %d\nstdout tail: %s\nstderr tail: %s" % (r.returncode, r.stdout[-1500:], r.stderr[-6000:])
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["rccl_ranks"] == 1 and line["dist_backend"] == "nccl" and line["config"]["hipgraph"] is True
    assert line["n_gpus"] == 1 and line["value"] > 0
