"""Mirror of the symmetric MoCo variant of `cet_pick/trains/tomo_moco_small_trainer.py` (`MoCoModel` :24-161,
`MoCoTrainer` :164-272): momentum update first, then the InfoNCE loss in both directions
(q(im1) against k(im2) and q(im2) against k(im1)), one enqueue of both key sets.

`_batch_shuffle_single_gpu` (:75-93) permutes the key batch so that BatchNorm statistics cannot leak between query and
key; batch statistics do not depend on the order of the rows, so on one device the shuffle changes nothing but the
floating-point summation order.  It is kept (a device-side randperm and two row gathers) for fidelity and can be
switched off (`shuffle=False`) for bit-reproducible tests.
"""
import torch
import torch.nn as nn

from .. import hipops as H
from ..models.moco import _world_size, batch_shuffle_ddp, batch_unshuffle_ddp, concat_all_gather
from .base_trainer import BaseTrainer


def _embed(enc, x):
    """Encoders here return [{'proj': z}]; the reference's small ResNet returns the embedding itself."""
    out = enc(x)
    if isinstance(out, (list, tuple)):
        out = out[0]
    if isinstance(out, dict):
        out = out["proj"]
    return out


class MoCoModel(nn.Module):
    def __init__(self, model_q, model_k, opt=None, dim=256, K=4096, m=0.99, T=0.1, bn_splits=8, symmetric=True,
                 shuffle=True):
        super().__init__()
        self.K, self.m, self.T, self.symmetric, self.shuffle, self.opt = K, m, T, symmetric, shuffle, opt
        self.encoder_q, self.encoder_k = model_q, model_k
        for param_q, param_k in zip(self.encoder_q.parameters(), self.encoder_k.parameters()):
            param_k.data.copy_(param_q.data)
            param_k.requires_grad = False
        self.register_buffer("queue", torch.randn(dim, K))
        self.queue = nn.functional.normalize(self.queue, dim=0)
        self.register_buffer("queue_ptr", torch.zeros(1, dtype=torch.long))
        self._arena_q = self._arena_k = None

    def flatten_parameters(self):
        if self._arena_q is None:
            self._arena_q, self._arena_k = H.ParamArena(self.encoder_q), H.ParamArena(self.encoder_k)
        return self._arena_q, self._arena_k

    @torch.no_grad()
    def _momentum_update_key_encoder(self):
        if self._arena_q is not None:
            H.ema_update_(self._arena_k.flat, self._arena_q.flat, self.m)
            return
        for param_q, param_k in zip(self.encoder_q.parameters(), self.encoder_k.parameters()):
            param_k.data.mul_(self.m).add_(param_q.data, alpha=1.0 - self.m)

    @torch.no_grad()
    def _dequeue_and_enqueue(self, keys):
        if self.K % keys.shape[0] != 0:
            raise AssertionError("K must be a multiple of the enqueued batch (tomo_moco_small_trainer.py:67)")
        H.queue_enqueue_(self.queue, self.queue_ptr, keys.contiguous())

    @torch.no_grad()
    def _batch_shuffle_single_gpu(self, x):
        idx_shuffle = torch.randperm(x.shape[0], device=x.device)
        return x[idx_shuffle], torch.argsort(idx_shuffle)

    @torch.no_grad()
    def _batch_unshuffle_single_gpu(self, x, idx_unshuffle):
        return x[idx_unshuffle]

    def contrastive_loss(self, im_q, im_k):
        q = H.l2_normalize(_embed(self.encoder_q, im_q))
        with torch.no_grad():
            if self.shuffle and _world_size() > 1:
                # shuffle-BN across the ranks (models/moco.py:55-99): every rank encodes a random share of the GLOBAL
                # key batch, so the key encoder's batch statistics are not those of the rank's own queries
                im_k_, idx_unshuffle = batch_shuffle_ddp(im_k)
                k = H.l2_normalize(_embed(self.encoder_k, im_k_))
                k = batch_unshuffle_ddp(k, idx_unshuffle)
            elif self.shuffle:
                im_k_, idx_unshuffle = self._batch_shuffle_single_gpu(im_k)
                k = H.l2_normalize(_embed(self.encoder_k, im_k_.contiguous()))
                k = self._batch_unshuffle_single_gpu(k, idx_unshuffle).contiguous()
            else:
                k = H.l2_normalize(_embed(self.encoder_k, im_k))
        logits = H.moco_logits(q, k, self.queue, self.T)          # (N, 1+K) = [q.k | q @ queue] / T
        loss = H.cross_entropy_label0(logits)
        return loss, q, k

    def forward(self, im1, im2):
        with torch.no_grad():
            self._momentum_update_key_encoder()
        if self.symmetric:
            loss_12, q1, k2 = self.contrastive_loss(im1, im2)
            loss_21, q2, k1 = self.contrastive_loss(im2, im1)
            loss = loss_12 + loss_21
            k = torch.cat([k1, k2], dim=0)
        else:
            loss, q, k = self.contrastive_loss(im1, im2)
        self._dequeue_and_enqueue(concat_all_gather(k) if _world_size() > 1 else k)
        return loss, {"loss": loss, "moco_loss": loss}


class _LossModule(nn.Module):
    """BaseTrainer's (model, loss) pair for a model that computes its own loss."""

    def __init__(self, model):
        super().__init__()
        self.model = model

    def forward(self, batch, epoch, phase):
        loss, loss_stats = self.model(batch["input"], batch["input_aug"])
        return None, loss, loss_stats


class MoCoTrainer(BaseTrainer):
    """tomo_moco_small_trainer.py:164-272: `model` is a MoCoModel; train / val return ({'loss','moco_loss','time'}, {})."""

    def __init__(self, opt, model, optimizer):
        self.opt, self.optimizer = opt, optimizer
        self.loss_stats = ["loss", "moco_loss"]
        self.model_with_loss = _LossModule(model)
        self.engine = None
        self.exchange = None
        self.device = None
        self.iter = 0

    def _make_engine(self):
        """No step engine (the model computes its own symmetric loss); under torch.distributed the query encoder's
        gradients are averaged over the ranks (the key encoder has none: it follows by EMA)."""
        self.engine = None
        if H._distributed():
            self.exchange = H.GradExchange(self.model_with_loss.model.encoder_q)
            self.exchange.broadcast_parameters(0)
