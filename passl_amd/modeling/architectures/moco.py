"""MoCo (v1/v2) on the MI355X HIP path.

Constructor, registry name, attribute names (``encoder_q``, ``encoder_k``, ``backbone``, ``head``,
buffers ``queue`` [dim,K] and ``queue_ptr`` int64[1]) and the ``forward(*inputs, mode=...)``
contract are the reference's (passl_v110/modeling/architectures/moco.py:25-195).  The step does
what moco.py:154-185 does, in this order:

  q = normalize(encoder_q(img_q))                         train-mode BN (batch statistics)
  no_grad: key-encoder EMA over ALL parameters incl. BN statistics (moco.py:82-90; the key
           encoder's BN runs on those running statistics, modules/freeze.py — SURVEY §3.1 note A)
           k = normalize(encoder_k(img_k))
  loss/acc = InfoNCE([q.k | q@queue]/T)                   fused, no [N,K+1] logits in HBM
  queue[:, ptr:ptr+B] = all_gather(k)^T ; ptr = (ptr+B) % K

Differences by design (output-neutral): (1) batch shuffle (moco.py:107-152) is off by default
because a key encoder with frozen-statistics BN is per-sample independent, so shuffling changes no
output; ``shuffle_bn=True`` restores the collective pattern.  (2) the EMA is one launch over a flat
buffer instead of 269 ``paddle.assign`` calls.  (3) encoders keep their state in EncoderArenas.
"""
import os

import torch
import torch.distributed as dist

from ...core.sync_utils import collectives_active
from ...hip import nn, ops, streams
from ...hip.nn import EncoderArena
from ...modules import freeze_batchnorm_statictis
from ..backbones import build_backbone
from ..heads import build_head
from ..necks import build_neck
from .builder import MODELS


def _world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


@torch.no_grad()
def concat_all_gather(tensor):
    """all_gather + concat along dim 0 (identity for a single process) — moco.py:198-210."""
    ws = _world_size()
    if not collectives_active() or 'nogather' in os.environ.get('PASSL_DP_DIAG', ''):
        return tensor
    out = torch.empty((ws * tensor.shape[0],) + tuple(tensor.shape[1:]), dtype=tensor.dtype,
                      device=tensor.device)
    src = tensor.contiguous()
    # (stays a live call when the step is replayed from a native plan: the plan is cut here, hip/replay.py)
    from ...hip.replay import host_call
    host_call(lambda: dist.all_gather_into_tensor(out, src))
    return out


@MODELS.register()
class MoCo(nn.Layer):
    # the step replays faithfully from a captured launch list (hip/graph.py, hip/replay.py): every step-varying
    # scalar lives on the device (learning rate, queue pointer), no random numbers are drawn inside the step —
    # tests/test_moco_gpu.py::test_step_graph_replay_is_bit_identical.  Models opt IN (Trainer._build_step_graph).
    graph_safe = True

    def __init__(self, backbone, neck=None, head=None, dim=128, K=65536, m=0.999, T=0.07,
                 shuffle_bn=False):
        super().__init__()
        self.K, self.m, self.T = K, m, T
        self.shuffle_bn = shuffle_bn
        self.encoder_q = torch.nn.Sequential(build_backbone(backbone), build_neck(neck))
        self.encoder_k = torch.nn.Sequential(build_backbone(backbone), build_neck(neck))
        self.backbone = self.encoder_q[0]
        self.head = build_head(head)
        # flat storage; key encoder = copy of the query encoder, no gradients, frozen-statistics BN
        self.arena_q = EncoderArena(self.encoder_q, trainable=True)
        self.arena_k = EncoderArena(self.encoder_k, trainable=False)
        self.arena_k.copy_from(self.arena_q)              # param_k.set_value(param_q), moco.py:69-72
        freeze_batchnorm_statictis(self.encoder_k)        # moco.py:74
        dev = self.arena_q.device
        queue = torch.randn(dim, K)                       # host RNG (seeded by the Trainer)
        queue = queue / queue.norm(dim=0, keepdim=True).clamp_min(1e-12)
        self.register_buffer('queue', queue.to(dev))
        self.register_buffer('queue_ptr', torch.zeros(1, dtype=torch.int64, device=dev))
        self._ptr = 0     # host mirror of queue_ptr: no device->host sync in the step
        self._enqueued = 0
        self._key_groups = None       # BatchNorm groups of the key pipeline (built on first use)

    # -- state ------------------------------------------------------------------------------
    def load_state_dict(self, state_dict, strict=True):
        r = super().load_state_dict(state_dict, strict=strict)
        self.sync_runtime_state()
        return r

    def sync_runtime_state(self):
        """Call after parameters/buffers were written from outside (checkpoint load, tests)."""
        self._ptr = int(self.queue_ptr[0].item())
        self.arena_q.refresh()
        self.arena_k.refresh()
        self.arena_k.update_bn_affine()

    # -- moco.py:82-90 ----------------------------------------------------------------------
    @torch.no_grad()
    def _momentum_update_key_encoder(self):
        self.arena_k.ema_from(self.arena_q, self.m)

    # -- moco.py:92-105 ---------------------------------------------------------------------
    @torch.no_grad()
    def _dequeue_and_enqueue(self, keys):
        keys = concat_all_gather(keys)
        batch_size = keys.shape[0]
        assert self.K % batch_size == 0  # for simplicity
        # the pointer lives on the device (the `queue_ptr` buffer itself): the kernel reads it, writes the keys and
        # advances it, so that a captured HIP graph of the step can be replayed; `_ptr` is the host's mirror of it
        # (kept without a device sync)
        ops.enqueue_dev(self.queue, keys.contiguous(), self.queue_ptr)
        self._enqueued = batch_size
        self._ptr = (self._ptr + batch_size) % self.K

    def on_graph_replay(self):
        """hip/graph.py: the captured step was replayed (no Python of train_iter ran): advance the host mirror."""
        self._ptr = (self._ptr + self._enqueued) % self.K

    # -- moco.py:107-152 (output-neutral here; see module docstring) -------------------------
    @torch.no_grad()
    def _batch_shuffle_ddp(self, x):
        bs = x.shape[0]
        x_gather = concat_all_gather(x)
        n_all = x_gather.shape[0]
        idx_shuffle = torch.randperm(n_all, device=x.device)
        if collectives_active():
            dist.broadcast(idx_shuffle, src=0)
        idx_unshuffle = torch.argsort(idx_shuffle)
        rank = dist.get_rank() if _world_size() > 1 else 0
        idx_this = idx_shuffle.view(n_all // bs, -1)[rank]
        return x_gather[idx_this], idx_unshuffle

    @torch.no_grad()
    def _batch_unshuffle_ddp(self, x, idx_unshuffle):
        bs = x.shape[0]
        x_gather = concat_all_gather(x)
        rank = dist.get_rank() if _world_size() > 1 else 0
        idx_this = idx_unshuffle.view(x_gather.shape[0] // bs, -1)[rank]
        return x_gather[idx_this]

    # -- key path under the query forward ---------------------------------------------------
    def _key_overlap(self, img):
        """May the key encoder run on its own stream, one trunk stage behind the query forward?"""
        if self._key_groups is None:
            bb = self.encoder_k[0]
            ok = all(hasattr(bb, n) for n in ('frozen_unit', 'units')) and \
                os.environ.get('PASSL_KEY_OVERLAP', '1') != '0'
            self._key_groups = False
            if ok:
                try:
                    self._key_groups = self.arena_k.bn_groups(bb.units())
                except AssertionError:
                    self._key_groups = False          # a neck with BatchNorm layers, an unusual trunk: keep the plain path
        return bool(self._key_groups) and not self.shuffle_bn and streams.enabled(img) and \
            not torch.cuda.is_current_stream_capturing()

    def _train_iter_overlapped(self, img_q, img_k):
        """Same arithmetic as train_iter's plain path, different schedule.  The momentum update
        (moco.py:82-90) couples the key encoder to THIS step's query forward only through the BatchNorm running
        statistics, layer by layer: the key encoder's unit u (ResNet.units(): one bottleneck) needs the statistics
        the query forward's unit u has just written.  So the parameter part of the update runs before the query
        forward, and each unit of the key encoder (statistics part of the update for that unit, folded affine, fused
        inference convs) is issued on the key stream right after the query encoder's same unit was enqueued — it
        executes under the query encoder's next units (MFMA-bound fused convs next to the query path's HBM-bound
        BatchNorm passes); only the last bottleneck and the neck remain behind the query forward."""
        dev = img_q.device
        main, key = torch.cuda.current_stream(dev), streams.key_stream(dev)
        bb_q, bb_k = self.encoder_q[0], self.encoder_k[0]
        with torch.no_grad():
            self.arena_k.ema_params_from(self.arena_q, self.m)
        state = {'y': img_k, 'hold': [img_k]}      # (a staged input lives in the SIDE stream's pool: keep it until the join)

        def unit_done(u):
            ev = streams.record_event(main)          # the query unit's statistics are final
            with torch.no_grad(), torch.cuda.stream(key):
                streams.wait_event(key, ev)
                self.arena_k.ema_stats_from(self.arena_q, self.m, self._key_groups[u])
                state['y'] = bb_k.frozen_unit(u, state['y'], allow_fork=False)
        bb_q._unit_done = unit_done
        try:
            q = self.encoder_q(img_q)               # queries: NxC (fp32)
        finally:
            bb_q._unit_done = None
        q = nn.normalize(q, axis=1)
        with torch.no_grad(), torch.cuda.stream(key):
            k = self.encoder_k[1](state['y'])
            k = nn.normalize(k, axis=1)
        streams.wait_stream(main, key)
        state.clear()                                # (inputs / activations of the key path: released after the join)
        return q, k

    # -- moco.py:154-185 --------------------------------------------------------------------
    def train_iter(self, *inputs, **kwargs):
        img_q, img_k = inputs
        self.arena_q.refresh()                      # compute-dtype copies of the updated weights
        if self._key_overlap(img_q):
            if hasattr(self.encoder_k[0], 'stage_input'):
                img_k = self.encoder_k[0].stage_input(img_k)
            q, k = self._train_iter_overlapped(img_q, img_k)
            with torch.no_grad():
                queue_snapshot = ops.clone(self.queue)     # `self.queue.clone().detach()`, moco.py:180
            outputs = self.head.fused(q, k, queue_snapshot)
            self._dequeue_and_enqueue(k)
            return outputs
        if not self.shuffle_bn and hasattr(self.encoder_k[0], 'stage_input'):
            # the key view's layout conversion does not depend on the EMA: side stream, under the query forward
            img_k = self.encoder_k[0].stage_input(img_k)

        q = self.encoder_q(img_q)                   # queries: NxC (fp32)
        q = nn.normalize(q, axis=1)
        # plain schedule (side stream off, shuffle_bn, graph capture): the whole momentum update after the query
        # forward — it covers the BatchNorm running statistics that forward has just updated (moco.py:82-90,
        # SURVEY §3.1 note A) — then the key forward
        with torch.no_grad():
            self._momentum_update_key_encoder()
            if self.shuffle_bn:
                img_k, idx_unshuffle = self._batch_shuffle_ddp(img_k)
            k = self.encoder_k(img_k)
            k = nn.normalize(k, axis=1)
            if self.shuffle_bn:
                k = self._batch_unshuffle_ddp(k, idx_unshuffle)
        with torch.no_grad():
            queue_snapshot = ops.clone(self.queue)     # `self.queue.clone().detach()`, moco.py:180
        outputs = self.head.fused(q, k, queue_snapshot)
        self._dequeue_and_enqueue(k)
        return outputs

    def forward(self, *inputs, mode='train', **kwargs):
        if mode == 'train':
            return self.train_iter(*inputs, **kwargs)
        elif mode == 'test':
            return self.test_iter(*inputs, **kwargs)
        elif mode == 'extract':
            with torch.no_grad():
                self.arena_q.refresh()
                return self.backbone(*inputs).permute(0, 3, 1, 2).float()
        else:
            raise Exception('No such mode: {}'.format(mode))
