"""One SimSiam training step as the reference's hot loop performs it (trains/base_trainer.py:486-508 ->
ModelWithLossSimSiam.forward :124-133 -> models/networks/simsiam_model_2d.py:776-819 two-view forward ->
trains/tomo_simsiam_trainer.py:28-40 loss -> zero_grad / backward / SGD step, simsiam_main.py:65), driven without per-iteration
host work and replayed from a hipGraph (round 6):

  * the model's parameters live in one flat fp32 arena (hipops.ParamArena) with TWO gradient arenas: both views run through the same
    weights, so every parameter receives two gradient contributions per step - the first is written into arena A, the second into
    arena B (no `grad.add_` launch per parameter, ~60 a step), and ONE fused kernel applies p -= lr (gA + gB) (mi_sgd_step2) instead of
    torch.optim.SGD's per-tensor loop;
  * the pre-cut weight images of the 3 x 3 layers (csrc/conv_p2d.hip: forward and data-gradient image per convolution) are kept here and
    re-cut by two launches behind the SGD kernel - not by one launch in front of each of the 48 direct-kernel calls;
  * the loss meters (loss, cosine_loss, output_std) are accumulated on the device by one launch inside the step and read once per
    print interval;
  * the whole step - two forwards, loss, backward, optimizer, image refresh, meters - is captured into a hipGraph on the third call and
    replayed from then on (RCCL collectives are captured with it; gloo ranks stay eager).

Data-parallel ranks exchange the two gradient arenas (bucketed all-reduce, RCCL over xGMI) between backward and the optimizer kernel,
whose grad_scale carries the 1 / world of DistributedDataParallel's averaging; SyncBN statistics are exchanged by the BatchNorm
modules themselves.
"""
import os
import warnings

import torch

from .. import hipops as H


def _dist():
    import torch.distributed as dist
    return dist if (dist.is_available() and dist.is_initialized()) else None


class SimSiamStepEngine:
    STAT_KEYS = ("loss", "cosine_loss", "output_std")

    def __init__(self, model_with_loss, lr, weight_decay=0.0, use_graph=False, n_buckets=2):
        self.mwl = model_with_loss
        self.model = model_with_loss.model
        self.lr, self.weight_decay = float(lr), float(weight_decay)
        self.arena = H.ParamArena(self.model, second_grad_arena=True)
        dev = self.arena.flat.device
        self.lr_dev = torch.full((1,), self.lr, dtype=torch.float32, device=dev)
        self._one = torch.ones((), dtype=torch.float32, device=dev)
        self.loss = torch.zeros((), dtype=torch.float32, device=dev)
        self.stat_sums = torch.zeros(4, dtype=torch.float32, device=dev)
        self.stats = {}
        d = _dist()
        self.world = d.get_world_size() if d else 1
        self.dist_on = H._distributed()
        graph_ok = (not self.dist_on) or (d.get_backend() == "nccl" and os.environ.get("CETPICK_DIST_GRAPH", "1") != "0")
        self.use_graph = bool(use_graph) and graph_ok and dev.type == "cuda"
        self._graph = None
        self._static = None
        self._calls = 0
        n = self.arena.numel
        per = (max((n + n_buckets - 1) // n_buckets, 1 << 18) + 3) // 4 * 4
        self.buckets = [(a, min(a + per, n)) for a in range(0, n, per)]
        self._images = self._build_images()
        self._img_versions = None

    # ---- pre-cut weight images of the 3 x 3 / stride-1 convolutions (conv_p2d.hip) -----------------------------------
    def _build_images(self):
        if not self.arena.flat.is_cuda:
            return None
        items, table = [], {}
        lib = H.L.lib()
        for m in self.model.modules():
            if isinstance(m, H.HipConv2d) and m.ci == m.co and m.k == 3 and m.stride == 1 and m.pad == 1 and m.co in (64, 128, 256) \
                    and H._phys_ok(m.weight):
                for dgrad in (0, 1):
                    img = torch.empty(int(lib.mi_conv2d_p2d_wimg_bytes(m.co)), dtype=torch.uint8, device=m.weight.device)
                    items.append((m.weight, dgrad, img))
                    table[(m.weight.data_ptr(), dgrad)] = img
        self._img_items = items
        return table

    def refresh_images(self):
        """Re-cut every image from the current weights (one launch per 16) on the current stream."""
        if self._images:
            H.p2d_prep(self._img_items)
            self._img_versions = self._weight_versions()

    def _weight_versions(self):
        return sum(it[0]._version for it in self._img_items[::2]) + self.arena.flat._version

    def broadcast_state(self, src=0):
        d = _dist()
        if d is None:
            return
        d.broadcast(self.arena.flat, src)
        for b in self.model.buffers():
            d.broadcast(b, src)
        self.refresh_images()

    def set_lr(self, lr):
        self.lr = float(lr)
        self.lr_dev.fill_(self.lr)

    # ---- the step ------------------------------------------------------------------------------------------------
    def _exchange(self):
        """Sum both gradient arenas over the ranks (the averaging rides in the optimizer kernel)."""
        for flat in (self.arena.flat_grad, self.arena.flat_grad2):
            for a, b in self.buckets:
                H.dist_all_reduce(flat[a:b])

    def _step_eager(self, batch):
        self.arena.zero_grad()
        H.ACTIVE_P2D = self._images
        try:
            _, loss, stats = self.mwl(batch, 0, "train")
            loss.backward(self._one)
        finally:
            H.ACTIVE_P2D = None
        self.arena.settle_grads()
        if self.dist_on:
            self._exchange()
        H.sgd_step2_(self.arena.flat, self.arena.flat_grad, self.arena.flat_grad2, self.lr, self.weight_decay, self.lr_dev,
                     grad_scale=1.0 / self.world)
        if self._images:
            H.p2d_prep(self._img_items)
        keys = [k for k in self.STAT_KEYS if k in stats]
        H.scalar_accumulate_(self.stat_sums, *[stats[k].detach().reshape(()) for k in keys])
        self._stat_keys = keys
        self.loss = loss.detach()
        self.stats = stats
        return self.loss

    def step_eager(self, x1, x2):
        return self._run({"input": x1, "input_aug": x2}, eager=True)

    def step(self, x1, x2):
        return self._run({"input": x1, "input_aug": x2})

    def step_batch(self, batch):
        return self._run({k: v for k, v in batch.items() if isinstance(v, torch.Tensor)})

    def take_stat_sums(self):
        """{key: sum over the steps since the last call} (one host read), and reset."""
        v = self.stat_sums.tolist()
        self.stat_sums.zero_()
        return dict(zip(getattr(self, "_stat_keys", self.STAT_KEYS), v))

    def _capture(self, batch):
        self._static = {k: v.clone() for k, v in batch.items()}
        torch.cuda.synchronize()
        if self.dist_on:
            from .moco_engine import MocoStepEngine
            MocoStepEngine._drain_watchdog()
        graph = torch.cuda.CUDAGraph(keep_graph=True)
        err = None
        try:
            with torch.cuda.graph(graph, capture_error_mode="thread_local" if self.dist_on else "global"):
                self._step_eager(self._static)
        except Exception as e:
            if not self.dist_on:
                raise
            err = e
        if self.dist_on:
            torch.cuda.synchronize()
            ok = torch.tensor([0 if err is not None else 1], dtype=torch.int32, device=self.lr_dev.device)
            _dist().all_reduce(ok, op=_dist().ReduceOp.MIN)
            if int(ok.item()) == 0:
                warnings.warn("hipGraph capture of the data-parallel SimSiam step failed (%s); every rank runs it eagerly" % (err,))
                self.use_graph, self._static = False, None
                return None
        self._graph_loss, self._graph_stats = self.loss, self.stats
        return graph

    def _run(self, batch, eager=False):
        if self._images and self._weight_versions() != self._img_versions:
            self.refresh_images()                     # first step, or the weights / the arena were written through torch ops
        if eager or not self.use_graph:
            return self._step_eager(batch)
        if self._graph is None:
            self._calls += 1
            if self._calls <= 2:
                return self._step_eager(batch)
            self._graph = self._capture(batch)
            if self._graph is None:
                return self._step_eager(batch)
        if any(k not in self._static or self._static[k].shape != v.shape for k, v in batch.items()):
            if self.dist_on:
                raise ValueError("data-parallel graph step: the batch differs from the captured one (use drop_last)")
            return self._step_eager(batch)
        for k, v in batch.items():
            self._static[k].copy_(v)
        H._bump_weight_epoch()                          # the replayed optimizer kernel writes the arena (no Python runs)
        self._graph.replay()
        self.loss, self.stats = self._graph_loss, self._graph_stats
        return self.loss

    def node_counts(self):
        if self._graph is None:
            return None
        import ctypes
        from .. import _lib as L
        counts = (ctypes.c_int * 4)()
        L.check(L.lib().mi_graph_node_counts(ctypes.c_void_p(self._graph.raw_cuda_graph()), ctypes.cast(counts, ctypes.c_void_p)),
                "mi_graph_node_counts")
        return dict(zip(("kernel", "memcpy", "memset", "other"), [int(c) for c in counts]))

    def close(self):
        if self._graph is not None:
            torch.cuda.synchronize()
            self._graph.reset()
            self._graph = None
        self._static = None
        self._calls = 0
        if torch.cuda.is_available():
            torch.cuda.synchronize()
