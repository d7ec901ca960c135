"""Mirror of cet_pick/models/moco.py (reference models/moco.py:12-162) on the MI355X kernels.

Same constructor, buffers (`queue`, `queue_ptr`) and `forward(im_q, im_k) -> (logits, labels)`.
Differences (reference defects): labels are created on the logits' device (the
reference hard-codes .cuda()), nothing is printed per step, the queue pointer never leaves the
device, and under torch.distributed the keys are all-gathered before the enqueue so every rank
holds the same queue (canonical MoCo; the reference leaves that to DDP's buffer broadcast).
"""
import torch
import torch.nn as nn

from .. import hipops as H


class MoCo(nn.Module):
    def __init__(self, encoder_q, encoder_k, dim=32, r=1024, m=0.999, T=0.1):
        super().__init__()
        self.r, self.m, self.T = r, m, T
        self.encoder_q = encoder_q
        self.encoder_k = encoder_k
        for param_q, param_k in zip(self.encoder_q.parameters(), self.encoder_k.parameters()):
            param_k.data.copy_(param_q.data)
            param_k.requires_grad = False
        self.register_buffer("queue", torch.randn(dim, r))
        self.queue = nn.functional.normalize(self.queue, dim=0)
        self.register_buffer("queue_ptr", torch.zeros(1, dtype=torch.long))
        self._arena_q = self._arena_k = None

    # ---- flat arenas (built once the model sits on the GPU) ------------------------------------
    def flatten_parameters(self):
        """Re-home both encoders' parameters in flat fp32 arenas: EMA, SGD and the gradient
        all-reduce then run as single passes.  Call after .cuda()/.to(device)."""
        if self._arena_q is None or self._arena_q.flat.device != next(self.encoder_q.parameters()).device:
            self._arena_q = H.ParamArena(self.encoder_q)
            self._arena_k = H.ParamArena(self.encoder_k)
            assert self._arena_q.numel == self._arena_k.numel
        return self._arena_q, self._arena_k

    @torch.no_grad()
    def _momentum_update_key_encoder(self):
        """models/moco.py:31-39: k <- m*k + (1-m)*q over parameters (buffers untouched)."""
        if self._arena_q is not None:
            H.ema_update_(self._arena_k.flat, self._arena_q.flat, self.m)
            if self.weight_images is not None and H.ACTIVE_IMAGES is self.weight_images:
                self.weight_images.refresh("k")        # encoder_k's pre-cut conv weights follow the update (one launch)
            return
        for param_q, param_k in zip(self.encoder_q.parameters(), self.encoder_k.parameters()):
            # un-flattened model: one launch per tensor over its (dense) storage
            n = param_q.numel()
            q = torch.as_strided(param_q.data, (n,), (1,))
            k = torch.as_strided(param_k.data, (n,), (1,))
            if n % 4 == 0 and k.data_ptr() % 16 == 0 and q.data_ptr() % 16 == 0:
                H.ema_update_(k, q, self.m)
            else:
                k.mul_(self.m).add_(q, alpha=1.0 - self.m)

    @torch.no_grad()
    def _dequeue_and_enqueue(self, keys):
        """models/moco.py:41-52, on the device (ptr is read and advanced by the kernel)."""
        H.queue_enqueue_(self.queue, self.queue_ptr, keys.contiguous())

    # The key branch (momentum update + encoder_k forward, no gradient) does not depend on the query branch: on one
    # GPU it runs on a second HIP stream next to encoder_q's forward.  The layer-2/3 launches of a batch-64 step are
    # too small to fill 256 CUs on their own, so the two forwards interleave on the chip.  Under torch.distributed
    # both branches issue SyncBN collectives from their own stream; every rank enqueues them in the same (program)
    # order, which is the order the process group's communicator stream runs them in.
    overlap_key_branch = True
    _side = None
    weight_images = None           # hipops.WeightImages of a MocoStepEngine (conv_direct3.hip), else None

    def _zero_labels(self, logits):
        """labels = zeros(B) (models/moco.py:140-141), allocated once per batch size instead of filled every step"""
        lab = getattr(self, "_labels", None)
        if lab is None or lab.shape[0] != logits.shape[0] or lab.device != logits.device:
            lab = torch.zeros(logits.shape[0], dtype=torch.long, device=logits.device)
            self._labels = lab
        return lab

    def _key_branch(self, im_k):
        with torch.no_grad():
            self._momentum_update_key_encoder()
            H.stamp("ema")
            k = self.encoder_k(im_k)[0]["proj"]
            return k                                       # un-normalised: the logits kernel normalises both branches

    # MocoStepEngine sets this: the keys are enqueued by flush_enqueue() AFTER the backward pass, so the backward reads the
    # queue in place (no 8 MB copy per step).  The state after the step is the same: the enqueue only writes the queue.
    defer_enqueue = False
    _pending_keys = None

    def flush_enqueue(self):
        if self._pending_keys is not None:
            self._dequeue_and_enqueue(self._pending_keys)
            self._pending_keys = None

    # Data parallel with SyncBN (moco_main.py:64-66): the two forward passes run LAYER-LOCKED - both encoders' forward generators
    # (forward_sync_gen) stop in front of each of their five statistics exchanges, and the sums of encoder_q's and encoder_k's layer i
    # go out as ONE collective (hipops.dist_all_reduce_pair: RCCL groups the two all-reduces into one launch, one xGMI latency) - 10
    # SyncBN collectives per step (5 forward pairs + 5 backward) instead of 15 (round 6).  Replicas stay identical: every
    # rank issues the same collectives in the same order.  OPT-IN (pair_sync_bn = True / CETPICK_PAIR_SYNCBN=1): on the one place it can be measured - the
    # 1-rank RCCL rehearsal, every collective issued and captured - the lock step costs more than it saves (1.627 against 1.580 ms per step, same box,
    # alternating runs): un-paired, a branch's exchange only holds ITS stream while the other branch computes; paired, both branches meet at every layer.
    # Whether five saved xGMI latencies outweigh that on real ranks is unmeasured (profiles/r06_experiments.txt item 9).
    import os as _os
    pair_sync_bn = _os.environ.get("CETPICK_PAIR_SYNCBN", "0") != "0"
    sync_collectives = 0           # collectives the last forward issued for SyncBN statistics (tests / diagnostics)

    def _paired_ok(self):
        def sync_all(enc):
            bns = [m for m in enc.modules() if isinstance(m, H.HipBatchNorm)]
            return bool(bns) and all(m.sync and (m.training or not m.track_running_stats) for m in bns)
        return (self.pair_sync_bn and H._distributed() and H.PROFILE is None and H.RELU_TAP is None
                and hasattr(self.encoder_q, "forward_sync_gen") and hasattr(self.encoder_k, "forward_sync_gen")
                and sync_all(self.encoder_q) and sync_all(self.encoder_k))

    def _forward_paired(self, im_q, im_k):
        """(q_raw, k_raw) with the branches' SyncBN exchanges paired.  On the GPU the key branch keeps its side stream: at a pair the
        main stream waits for the key branch's sums, all-reduces both, and the side stream waits for the result."""
        two = self.overlap_key_branch and im_q.is_cuda
        cur = side = None
        if two:
            if self._side is None:
                self._side = torch.cuda.Stream(device=im_q.device)
            cur, side = torch.cuda.current_stream(), self._side
            side.wait_stream(cur)

        class _Null:
            def __enter__(self): return None
            def __exit__(self, *a): return False
        on_side = (lambda: torch.cuda.stream(side)) if two else _Null
        with on_side(), torch.no_grad():
            self._momentum_update_key_encoder()
        gq, gk = self.encoder_q.forward_sync_gen(im_q), self.encoder_k.forward_sync_gen(im_k)
        q_out = k_out = None
        n_coll = 0
        while True:
            sq = sk = None
            try:
                sq = next(gq)
            except StopIteration as e:
                q_out = e.value
            with on_side(), torch.no_grad():
                try:
                    sk = next(gk)
                except StopIteration as e:
                    k_out = e.value
            if (sq is None) != (sk is None):
                raise RuntimeError("the two encoders stopped at different SyncBN layers: they are not the same architecture")
            if sq is None:
                break
            if two:
                ev = torch.cuda.Event()
                ev.record(side)
                cur.wait_event(ev)
                sk.record_stream(cur)
            H.dist_all_reduce_pair(sq, sk)
            n_coll += 1
            if two:
                ev2 = torch.cuda.Event()
                ev2.record(cur)
                side.wait_event(ev2)
        self.sync_collectives = n_coll
        q_raw, k_raw = q_out[0]["proj"], k_out[0]["proj"].detach()
        if two:
            cur.wait_stream(side)
            k_raw.record_stream(cur)
        return q_raw, k_raw

    def forward(self, im_q, im_k):
        if self._paired_ok():
            q_raw, k_raw = self._forward_paired(im_q, im_k)
        elif self.overlap_key_branch and im_q.is_cuda:
            if self._side is None:
                self._side = torch.cuda.Stream(device=im_q.device)
            cur = torch.cuda.current_stream()
            self._side.wait_stream(cur)                    # inputs and last step's SGD are ordered before it
            with torch.cuda.stream(self._side):
                H.STAMP_TAG = "k:"
                H.stamp("start")
                k_raw = self._key_branch(im_k)
                H.stamp("end")
            H.STAMP_TAG = "q:"
            H.stamp("start")
            q_raw = self.encoder_q(im_q)[0]["proj"]
            H.stamp("end")
            H.STAMP_TAG = ""
            cur.wait_stream(self._side)
            H.stamp("joined")
            k_raw.record_stream(cur)
        else:
            q_raw = self.encoder_q(im_q)[0]["proj"]
            k_raw = self._key_branch(im_k)
        # normalize(q), normalize(k) and the logits in one launch; k comes back normalised for the queue
        logits, k = H.moco_logits_normalized(q_raw, k_raw, self.queue, self.T, queue_stable=self.defer_enqueue)
        labels = self._zero_labels(logits)
        keys = concat_all_gather(k) if H._distributed() else k
        if self.defer_enqueue:
            self._pending_keys = keys
        else:
            self._dequeue_and_enqueue(keys)
        return logits, labels


def _world_size():
    import torch.distributed as dist
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


@torch.no_grad()
def concat_all_gather(tensor):
    """models/moco.py:149-162: all_gather (RCCL) + cat along the batch; no gradient."""
    import torch.distributed as dist
    world = dist.get_world_size()
    t = tensor.contiguous()
    if t.is_cuda and dist.get_backend() == "nccl":
        # ONE collective straight into the concatenated result: the list form leaves RCCL's flat receive buffer through one
        # device copy per rank and a cat - three memcpy nodes in the captured N > 1 step
        out = torch.empty((world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        H.dist_all_gather_into(out, t)
        return out
    tensors_gather = [torch.empty_like(t) for _ in range(world)]
    H.dist_all_gather(tensors_gather, t)
    return torch.cat(tensors_gather, dim=0)


@torch.no_grad()
def batch_shuffle_ddp(x):
    """models/moco.py:55-82 `_batch_shuffle_ddp` (shuffle-BN across ranks): all-gather the key batch, draw ONE
    permutation of the global batch (rank 0's, broadcast), keep this rank's share of the shuffled batch.
    Returns (x_this, idx_unshuffle).  The permutation is drawn on the device the batch lives on."""
    import torch.distributed as dist
    batch_size_this = x.shape[0]
    x_gather = concat_all_gather(x)
    batch_size_all = x_gather.shape[0]
    num_gpus = batch_size_all // batch_size_this
    idx_shuffle = torch.randperm(batch_size_all, device=x.device)
    dist.broadcast(idx_shuffle, src=0)
    idx_unshuffle = torch.argsort(idx_shuffle)
    idx_this = idx_shuffle.view(num_gpus, -1)[dist.get_rank()]
    return x_gather[idx_this].contiguous(), idx_unshuffle


@torch.no_grad()
def batch_unshuffle_ddp(x, idx_unshuffle):
    """models/moco.py:84-99 `_batch_unshuffle_ddp`: gather the keys of the shuffled shares and take back this rank's
    rows in their original order."""
    import torch.distributed as dist
    batch_size_this = x.shape[0]
    x_gather = concat_all_gather(x)
    num_gpus = x_gather.shape[0] // batch_size_this
    idx_this = idx_unshuffle.view(num_gpus, -1)[dist.get_rank()]
    return x_gather[idx_this].contiguous()
