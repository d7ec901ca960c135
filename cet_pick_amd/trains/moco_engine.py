"""One MoCo training step as the reference's hot loop performs it
(trains/base_trainer.py:486-508 -> models/moco.py:101-146 -> trains/tomo_moco_trainer.py:73 ->
optimizer.step), driven without per-iteration host syncs and replayed from a hipGraph - on one GPU and, with the RCCL
backend, on N GPUs too (the collectives are captured with the kernels; CETPICK_DIST_GRAPH=0 keeps the N>1 step eager,
and a capture that fails falls back to the eager step by itself).

Data-parallel ranks (one process per GPU, torch.distributed backend "nccl" = RCCL) exchange per step:
gradient arena all-reduce (40.4 MB fp32), MoCo key all-gather (B x 128), SyncBN per-channel sums.
"""
import os
import warnings

import torch

from .. import hipops as H


def _dist():
    import torch.distributed as dist
    return dist if (dist.is_available() and dist.is_initialized()) else None


class MocoStepEngine:
    def __init__(self, moco, lr, weight_decay=0.0, use_graph=False):
        self.moco = moco
        self.lr = float(lr)
        self.weight_decay = float(weight_decay)
        self.arena_q, self.arena_k = moco.flatten_parameters()
        dev = self.arena_q.flat.device
        self.lr_dev = torch.full((1,), self.lr, dtype=torch.float32, device=dev)
        self.logits = None
        self._one = torch.ones((), dtype=torch.float32, device=dev)
        self.loss = torch.zeros((), dtype=torch.float32, device=dev)
        self._loss_buf = self.loss
        # running sum of the steps' losses since take_loss_sum(): accumulated by one launch INSIDE the step (a graph node) - run_epoch's
        # meters read it at print time instead of launching a mean and an add behind every step
        self.loss_sum = torch.zeros((), dtype=torch.float32, device=dev)
        d = _dist()
        self.world = d.get_world_size() if d else 1
        self.dist_on = H._distributed()
        # an eager N>1 step is launch-bound on the host (3.5 ms against 2.4 ms on one GPU, before any collective); RCCL
        # collectives can be captured into the graph, gloo's (host-side) cannot
        graph_ok = (not self.dist_on) or (d.get_backend() == "nccl" and os.environ.get("CETPICK_DIST_GRAPH", "1") != "0")
        self.use_graph = bool(use_graph) and graph_ok
        self._graph = None
        self._static_q = self._static_k = None
        self._xchg = None                # side stream of the gradient exchange (overlaps the backward pass)
        self.buckets_sent = []           # tags of the last step's exchanges, in issue order (tests / diagnostics)
        if self.dist_on:
            self._setup_buckets()
        # weight gradients of layer1-3 and the head on a side stream, one fork per stage (hipops.SIDE_WGRADS)
        self.side_wgrads = dev.type == "cuda" and os.environ.get("CETPICK_SIDE_WGRADS", "1") != "0"
        self._wside = None
        self._wside_used = False
        self._wside_keep = []
        if self.side_wgrads:
            moco.encoder_q.grad_marker = self._on_marker
        self._images = self._build_weight_images()
        self._img_versions = None

    # ---- pre-cut weight images of the layer1 convolutions (conv_direct3.hip) ---------------------------------
    def _build_weight_images(self):
        """Group "q": forward + data-gradient images of encoder_q's 64 -> 64 3^3 convolutions (re-cut behind the SGD
        kernel), group "k": forward images of encoder_k's (re-cut behind the EMA kernel) - one launch each per step
        instead of one image build in front of each of the 12 direct-kernel launches."""
        if not self.arena_q.flat.is_cuda:
            return None
        imgs = H.WeightImages()
        for enc, group in ((self.moco.encoder_q, "q"), (self.moco.encoder_k, "k")):
            for m in enc.modules():
                if isinstance(m, H.HipConv3d) and H.WeightImages.eligible(m.weight, m.k, m.stride, m.pad):
                    imgs.add(group, m.weight, False)
                    if group == "q":
                        imgs.add(group, m.weight, True)
            # the stride-2 block fronts (conv_s2.hip): forward image per encoder, data-gradient image for encoder_q
            for blk in enc.modules():
                ds = getattr(blk, "downsample", None)
                if ds is None or getattr(blk, "stride", 1) != 2 or not hasattr(blk, "conv1"):
                    continue
                w, wds = blk.conv1.weight, ds[0].weight
                co, ci = int(w.shape[0]), int(w.shape[1])
                if (co, ci) not in ((128, 64), (256, 128)) or not (H._phys_ok(w) and H._phys_ok(wds)):
                    continue
                imgs.add_s2(group, w, wds, False)
                if group == "q":
                    imgs.add_s2(group, w, wds, True)
        self.moco.weight_images = imgs                 # MoCo re-cuts group "k" right behind its momentum update
        return imgs

    def refresh_weight_images(self):
        """Call after writing the encoders' weights from outside the step (checkpoint load, broadcast): the step itself
        keeps the images current, and notices writes made through torch ops by their version counters."""
        if self._images is not None:
            self._images.refresh("q")
            self._images.refresh("k")
            self._img_versions = self._weight_versions()

    def _weight_versions(self):
        """Changes whenever a torch op wrote a cached weight OR either flat arena (dist.broadcast, arena.flat.copy_, a
        checkpoint load): the engine's own SGD / EMA kernels go through the C-ABI, bump nothing, and refresh by themselves."""
        return self._images.versions() + self.arena_q.flat._version + self.arena_k.flat._version

    # ---- data parallel: bucketed gradient all-reduce overlapped with the backward pass ---------------------
    def _setup_buckets(self):
        """Arena ranges whose gradients are complete at each stage boundary of the backward pass (parameters sit in
        the arena in registration order: stem, layer1, layer2, layer3, feature_3d, fc, heads)."""
        enc = self.moco.encoder_q
        first = {}
        for (name, _), off in zip(enc.named_parameters(), self.arena_q.offsets):
            first.setdefault(name.split(".")[0], off)
        end = self.arena_q.numel
        l1, l2, l3 = first["layer1"], first["layer2"], first["layer3"]
        # marker tag -> range that is final when the gradient of that stage's INPUT exists
        self._bucket = {"layer3": (l3, end), "layer2": (l2, l3), "layer1": (l1, l2), "stem": (0, l1)}
        enc.grad_marker = self._on_marker

    def _reduce_bucket(self, tag):
        """One bucket of the gradient arena goes out on the exchange stream: forked from the stream the backward pass is
        on (this runs in an autograd hook, i.e. right behind the last kernel that wrote the bucket), joined again before
        the optimizer step.  The collective itself is issued synchronously - its internal wait only holds the exchange
        stream - because an async Work under hipGraph capture kills the process-group watchdog (hipops.dist_all_reduce)."""
        a, b = self._bucket[tag]
        H.flush_wgrad_reduces()                       # the bucket's weight gradients still sit in split-K slabs
        # (with the weight gradients on their side stream the slabs of the stage were reduced there, behind the launches
        # that wrote them: the exchange stream then waits for both)
        if b > a and not self.arena_q.flat_grad.is_cuda:        # (CPU tensors over gloo: plumbing tests)
            H.dist_all_reduce(self.arena_q.flat_grad[a:b])
            self.buckets_sent.append(tag)
        elif b > a:
            cur = torch.cuda.current_stream()
            if self._xchg is None:
                self._xchg = torch.cuda.Stream(device=self.arena_q.flat_grad.device)
            self._xchg.wait_stream(cur)
            if self._wside_used:
                self._xchg.wait_stream(self._wside)
            with torch.cuda.stream(self._xchg):
                H.dist_all_reduce(self.arena_q.flat_grad[a:b])
            self.buckets_sent.append(tag)

    # CETPICK_L2_WGRAD_LATE=1 (measured, not faster: r05_experiments.txt item 9): layer2's weight gradients wait for layer1's marker and
    # run next to the stem chain instead of next to layer1's data-gradient chain; the layer2 bucket of the gradient exchange then goes
    # out with layer1's.
    l2_late = os.environ.get("CETPICK_L2_WGRAD_LATE", "0") != "0"

    def _on_marker(self, tag):
        late = self.l2_late and self.side_wgrads
        if self.side_wgrads and tag in ("layer3", "layer2", "layer1") and not (late and tag == "layer2"):
            self._issue_side_wgrads(enqueue=(tag == "layer3"))
        if self.dist_on:
            if late and tag == "layer2":
                return
            if late and tag == "layer1":
                self._reduce_bucket("layer2")
            self._reduce_bucket(tag)

    def _issue_side_wgrads(self, enqueue=False):
        """The data-gradient chain has left a stage: its collected weight-gradient launches go out on the side stream (one
        fork), next to the following stage's kernels on the main stream.  The deferred key enqueue rides along behind
        layer3's: the backward pass read the queue for the last time in the logits' gradient, long before."""
        items = H.SIDE_WGRADS
        if not items:
            return
        cur = torch.cuda.current_stream()
        if H.PROFILE is not None:                      # bench.py's roofline pass: every call timed by itself, in line
            H.run_wgrad_jobs(items)
            H.flush_wgrad_reduces()
            if enqueue:
                self.moco.flush_enqueue()
            del items[:]
            return
        if self._wside is None:
            self._wside = torch.cuda.Stream(device=self.arena_q.flat_grad.device)
        self._wside.wait_stream(cur)
        with torch.cuda.stream(self._wside):
            H.run_wgrad_jobs(items)                   # (convolutions of one geometry: one launch for the group)
            H.flush_wgrad_reduces()                   # the stage's split-K slabs, behind the launches that wrote them
            if enqueue:
                self._wside_keep.append(self.moco._pending_keys)
                self.moco.flush_enqueue()
        # the launches read activations / gradients allocated on the main stream: they stay referenced until the join, so
        # that the allocator cannot hand their memory to the main stream's next kernels while the side stream reads it
        self._wside_keep.extend(items)
        del items[:]
        self._wside_used = True

    def broadcast_state(self, src=0):
        """Identical replicas before the first step (what DistributedDataParallel does at construction): the two flat
        parameter arenas (the parameters themselves are kernel-layout views, which RCCL refuses as non-contiguous), the
        buffers and the queue."""
        d = _dist()
        if d is None:
            return
        d.broadcast(self.arena_q.flat, src)
        d.broadcast(self.arena_k.flat, src)
        for b in self.moco.buffers():
            d.broadcast(b, src)
        self.refresh_weight_images()

    def set_lr(self, lr):
        """utils/utils.py:58-70 `adjust_learning_rate` target: the schedule reaches a captured graph
        through a device scalar."""
        self.lr = float(lr)
        self.lr_dev.fill_(self.lr)

    def _step_eager(self, im_q, im_k):
        moco = self.moco
        self.buckets_sent = []
        self.arena_q.zero_grad()
        H.ACTIVE_IMAGES = self._images                 # the cached weight images are valid inside the step only
        moco.defer_enqueue = True                      # the backward reads the queue in place; keys go in behind it
        try:
            H.stamp("step:start")
            logits, labels = moco(im_q, im_k)
            self.logits = logits.detach()              # (B, 1 + r) of the last step; under graph replay a static buffer
            loss = H.cross_entropy_label0(logits, out=self._loss_buf)      # lands in the engine's loss buffer: no copy
            H.DEFERRED_WGRADS = [] if self.arena_q.flat_grad.is_cuda else None     # split-K slabs of the wgrads: one reduce
            H.SIDE_WGRADS = [] if self.side_wgrads else None
            self._wside_used = False
            loss.backward(self._one)                   # (a kept seed: autograd's ones_like(loss) is a fill launch per step)
            if H.SIDE_WGRADS:                          # collected behind the last stage boundary: in line
                H.run_wgrad_jobs(H.SIDE_WGRADS)
            H.SIDE_WGRADS = None
            if self._wside_used:
                torch.cuda.current_stream().wait_stream(self._wside)
            del self._wside_keep[:]
            H.stamp("backward:end")
            H.flush_wgrad_reduces()
            moco.flush_enqueue()
        finally:
            H.DEFERRED_WGRADS = None
            H.SIDE_WGRADS = None
            H.ACTIVE_IMAGES = None
            moco.defer_enqueue = False
            moco._pending_keys = None
        if self.dist_on:
            # layer3+heads, layer2 and layer1 went out from the autograd hooks while the backward was still running
            # (RCCL over xGMI on its own stream); the stem's gradients are the last to exist
            self._reduce_bucket("stem")
            if self._xchg is not None:
                torch.cuda.current_stream().wait_stream(self._xchg)
        # (flat_grad holds the SUM over the ranks; the 1 / world of DistributedDataParallel's averaging rides in the SGD
        # kernel instead of a pass of its own over the arena)
        H.sgd_step_(self.arena_q.flat, self.arena_q.flat_grad, self.lr, self.weight_decay, self.lr_dev,
                    grad_scale=1.0 / self.world)
        if self._images is not None:
            self._images.refresh("q")                  # next step's forward / data-gradient images of encoder_q
        self.loss_sum.add_(self.loss)
        H.stamp("step:end")
        return self.loss

    def take_loss_sum(self):
        """Sum of the losses of the steps since the last call (one host sync), and reset."""
        v = float(self.loss_sum.item())
        self.loss_sum.zero_()
        return v

    @staticmethod
    def _drain_watchdog():
        """Data parallel, before a capture: wait until the process group's watchdog thread holds no Work of the eager steps.
        The watchdog polls its list every 100 ms (hipEventQuery on each Work's end event) and drops the Works it finds
        complete.  A Work of the eager warm-up steps that is still on that list when the capture starts gets polled DURING
        the capture - and ProcessGroupNCCL's internal communication stream, on which that end event was recorded, is by then
        part of the capture: ROCm answers hipErrorCapturedEvent ("operation not permitted on an event last recorded in a
        capturing stream") for an event whose stream is capturing NOW, the watchdog rethrows and the process aborts
        (profiles/r04_watchdog_abort.txt: 1 run in ~10; `thread_local` capture mode cured the other form of this race, the
        query of an unrelated event under `global` mode).
        The drain is a synchronisation, not a timer: the caller has synchronised the device (every Work is complete), and
        `ProcessGroup._wait_for_pending_works()` (c10d: ProcessGroupNCCL::waitForPendingWorks) returns once it has seen, under
        the watchdog's own two mutexes, BOTH the watchdog's work list and its completed-work list empty - it re-checks every
        watchdog poll period until then.  Nothing is issued between that return and the capture, so the list is still empty
        when the capture begins, and every collective of the captured step is synchronous (no Work is registered under
        capture).  Paid once per capture, never per step.  Only a torch build without the binding falls back to waiting
        three poll periods (CETPICK_WATCHDOG_DRAIN_S, default 0.3 s) - and says so."""
        d = _dist()
        pg = d.distributed_c10d._get_default_group()
        wait = getattr(pg, "_wait_for_pending_works", None)
        if wait is not None:
            wait()
            return
        import os
        import time
        warnings.warn("this torch has no ProcessGroup._wait_for_pending_works: draining the watchdog by a timed wait")
        time.sleep(float(os.environ.get("CETPICK_WATCHDOG_DRAIN_S", "0.3")))

    def _capture(self, im_q, im_k):
        """Record one step into a hipGraph.  Returns the graph, or None when the data-parallel ranks agreed to stay
        eager.  With collectives in the step every rank must take the same decision: a rank that replays a graph
        and a rank that launches eagerly no longer issue their collectives in one order."""
        self._static_q = im_q.clone()
        self._static_k = im_k.clone()
        torch.cuda.synchronize()
        if self.dist_on:
            self._drain_watchdog()
        graph = torch.cuda.CUDAGraph(keep_graph=True)      # the hipGraph_t stays queryable (node_counts)
        err = None
        try:
            # Data parallel: the process group's watchdog THREAD polls (hipEventQuery) Works at its own pace.  Two races, two
            # cures: (1) under the default "global" capture mode ANY such call from another thread while this one is
            # capturing terminates the process (hipErrorStreamCaptureUnsupported) - "thread_local" restricts only the capturing
            # thread; (2) a query of an event whose own stream has joined the capture fails in either mode
            # (hipErrorCapturedEvent) - _drain_watchdog() above emptied the watchdog's list, and nothing captured adds to it.
            mode = "thread_local" if self.dist_on else "global"
            # (the capture runs on a stream of its own: the kept-clean workspaces the eager steps made for THEIR stream get a twin for it
            # now, or their zero-fill would be recorded and replay with every step)
            cap = torch.cuda.Stream(device=self.lr_dev.device)
            if os.environ.get("CETPICK_PRIME_WS", "1") != "0":            # (A/B: 0 leaves the fill in the graph)
                H.L.prime_workspaces_for_stream(torch.cuda.current_stream(), cap)
            with torch.cuda.graph(graph, stream=cap, capture_error_mode=mode):       # records, does not execute
                self._step_eager(self._static_q, self._static_k)
        except Exception as e:                        # e.g. a collective that cannot be captured
            if not self.dist_on:
                raise
            err = e
        # the graph bakes in the addresses of the weight-gradient slab buffers: they must never be reallocated from now on
        for prm in self.arena_q.params:
            if getattr(prm, "_mi_slabs", None) is not None:
                prm._mi_slabs_pinned = True
        if self.dist_on:
            # the outcome is agreed on eagerly (outside any capture); a stream or communicator left in an error
            # state by the aborted capture surfaces here instead of being swallowed
            torch.cuda.synchronize()
            ok = torch.tensor([0 if err is not None else 1], dtype=torch.int32, device=self.lr_dev.device)
            _dist().all_reduce(ok, op=_dist().ReduceOp.MIN)
            if int(ok.item()) == 0:
                warnings.warn("hipGraph capture of the data-parallel step failed on %s (%s); every rank runs it eagerly"
                              % ("this rank" if err is not None else "another rank", err))
                del graph
                self.use_graph = False
                self._static_q = self._static_k = None
                return None
        return graph

    def step(self, im_q, im_k):
        """Returns the loss as a 0-d device tensor (no host sync).

        Graph mode: the first two calls run eagerly (they size every workspace), the third call
        captures the step into a hipGraph and from then on each call is one graph replay.  A batch whose
        shape differs from the captured one (a short last batch) runs eagerly."""
        if self._images is not None and self._weight_versions() != self._img_versions:
            self.refresh_weight_images()               # first step, or the weights / arenas were written through torch ops
        if not self.use_graph:
            return self._step_eager(im_q, im_k)
        if self._graph is None:
            self._calls = getattr(self, "_calls", 0) + 1
            if self._calls <= 2:
                return self._step_eager(im_q, im_k)
            self._graph = self._capture(im_q, im_k)
            if self._graph is None:
                return self._step_eager(im_q, im_k)
        if im_q.shape != self._static_q.shape or im_k.shape != self._static_k.shape:
            if self.dist_on:
                raise ValueError("data-parallel graph step: batch %s differs from the captured %s (use drop_last)"
                                 % (tuple(im_q.shape), tuple(self._static_q.shape)))
            return self._step_eager(im_q, im_k)
        if os.environ.get("CETPICK_COPY_PAIR", "1") != "0":
            H.copy_pair_(self._static_q, im_q, self._static_k, im_k)      # (one launch for both views)
        else:
            self._static_q.copy_(im_q)
            self._static_k.copy_(im_k)
        H._bump_weight_epoch()                          # the replayed SGD / momentum kernels write the arenas (no Python runs)
        self._graph.replay()
        return self.loss

    def node_counts(self):
        """{'kernel', 'memcpy', 'memset', 'other'} nodes of the captured step (None while the step runs eagerly)."""
        if self._graph is None:
            return None
        import ctypes
        from .. import _lib as L
        counts = (ctypes.c_int * 4)()
        L.check(L.lib().mi_graph_node_counts(ctypes.c_void_p(self._graph.raw_cuda_graph()), ctypes.cast(counts, ctypes.c_void_p)),
                "mi_graph_node_counts")
        return dict(zip(("kernel", "memcpy", "memset", "other"), [int(c) for c in counts]))

    def close(self):
        """Release everything that refers to the process group's communicator BEFORE the group is destroyed: the
        captured hipGraph holds the RCCL kernels of its collectives, so it has to go first; then the device is drained.
        Call before dist.destroy_process_group()."""
        if self._graph is not None:
            torch.cuda.synchronize()
            self._graph.reset()
            self._graph = None
        self._static_q = self._static_k = None
        self._calls = 0
        torch.cuda.synchronize()
