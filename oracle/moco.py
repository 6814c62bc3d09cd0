"""Oracle: one MoCo-v2 training step, torch-CPU fp32 (+ numpy fp64 head).

TEST INFRASTRUCTURE (see oracle/__init__.py).  Follows:

* passl_v110/modeling/architectures/moco.py:33-80   (__init__: q/k encoders,
  q->k copy, frozen-BN key encoder, queue = normalize(randn[dim,K], axis=0))
* moco.py:82-90    _momentum_update_key_encoder over *all* parameters()
  — including BN running stats, which in Paddle are non-trainable parameters
  (SURVEY §3.1 note A)
* moco.py:92-105   _dequeue_and_enqueue
* moco.py:154-185  train_iter (batch shuffle skipped: output-neutral because
  the key encoder's BN uses global stats, note A; and moco.py:121 needs CUDA)
* passl_v110/modeling/heads/contrastive_head.py:37-78  (InfoNCE + accuracy)
* passl_v110/hooks/optimizer_hook.py:25-50  clear_grad / backward / step
* paddle.optimizer.Momentum with float weight_decay = L2Decay folded into the
  gradient; update rule as restated in-tree at
  passl/optimizer/momentum.py:150-158:  g += wd*p; v = mu*v + g; p -= lr*v
* passl_v110/solver/builder.py:26-30 + paddle CosineAnnealingDecay (closed form)
"""
import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

from . import resnet50 as R


def l2_normalize(x, axis, eps=1e-12):
    """paddle.nn.functional.normalize: x / max(||x||_2, eps)  [Paddle-semantics]."""
    n = x.pow(2).sum(dim=axis, keepdim=True).sqrt().clamp_min(eps)
    return x / n


def contrastive_head(pos, neg, temperature):
    """ContrastiveHead.forward (contrastive_head.py:37-60) in torch fp32.
    Returns loss [1]-like scalar tensor, acc1, acc5 (percent), logits."""
    N = pos.shape[0]
    logits = torch.cat((pos, neg), dim=1) / temperature
    labels = torch.zeros(N, dtype=torch.int64)
    loss = F.cross_entropy(logits, labels)                 # mean reduction
    with torch.no_grad():
        _, pred = logits.topk(5, 1, True, True)            # contrastive_head.py:69
        correct = (pred.t() == labels.reshape(1, -1)).float()
        acc1 = correct[:1].reshape(-1).sum() * 100.0 / N
        acc5 = correct[:5].reshape(-1).sum() * 100.0 / N
    return loss, acc1, acc5, logits


def contrastive_head_f64(pos, neg, temperature):
    """Same head in numpy float64 (spot check for the fp32 paths)."""
    pos = np.asarray(pos, dtype=np.float64)
    neg = np.asarray(neg, dtype=np.float64)
    logits = np.concatenate([pos, neg], axis=1) / temperature
    m = logits.max(axis=1, keepdims=True)
    lse = m[:, 0] + np.log(np.exp(logits - m).sum(axis=1))
    loss = float((lse - logits[:, 0]).mean())
    rank = (logits[:, 1:] > logits[:, :1]).sum(axis=1)     # strictly-greater count
    acc1 = float((rank < 1).mean() * 100.0)
    acc5 = float((rank < 5).mean() * 100.0)
    return loss, acc1, acc5, logits


def cosine_lr(base_lr, step, t_max, eta_min=0.0):
    """paddle.optimizer.lr.CosineAnnealingDecay closed form at epoch=step."""
    return eta_min + (base_lr - eta_min) * (1 + math.cos(math.pi * step / t_max)) / 2


class MoCoOracle:
    """State + step of MoCo (moco.py) with Momentum-SGD (optimizer_hook.py)."""

    def __init__(self, dim=128, K=65536, m=0.999, T=0.2, lr=0.015, t_max=200 * 5004,
                 weight_decay=1e-4, momentum=0.9, seed=0, width_div=1, neck='NonLinearNeckV1',
                 milestones=None, bf16=False):
        # neck='LinearNeck', T=0.07, lr=0.03, milestones=[120, 160] epochs = configs/moco/moco_v1_r50.yaml
        # bf16=True: bf16-emulating encoders (oracle/bf16.py); head, EMA and optimizer stay fp32 as
        # in the product path
        self.milestones = milestones
        self.bf16 = bf16
        gen = torch.Generator().manual_seed(seed)
        self.K, self.m, self.T = K, m, T
        self.base_lr, self.t_max = lr, t_max
        self.wd, self.mu = weight_decay, momentum
        self.q = R.init_encoder_state(gen, out_channels=dim, width_div=width_div, neck=neck)
        # moco.py:69-72  param_k.set_value(param_q) for every parameter (incl. BN stats)
        self.k = OrderedDict((n, t.clone()) for n, t in self.q.items())
        # moco.py:77-80
        self.queue = l2_normalize(torch.randn(dim, K, generator=gen), axis=0)
        self.queue_ptr = 0
        self.velocity = OrderedDict()
        self.step_count = 0

    # -- moco.py:82-90 -------------------------------------------------------
    @torch.no_grad()
    def momentum_update_key_encoder(self):
        for n in self.q:
            self.k[n] = self.k[n] * self.m + self.q[n] * (1.0 - self.m)

    # -- moco.py:92-105 ------------------------------------------------------
    @torch.no_grad()
    def dequeue_and_enqueue(self, keys):
        bs = keys.shape[0]
        assert self.K % bs == 0
        ptr = self.queue_ptr
        self.queue[:, ptr:ptr + bs] = keys.t()
        self.queue_ptr = (ptr + bs) % self.K

    def lr(self):
        if self.milestones is not None:       # MultiStepDecay (milestones already in iterations), gamma 0.1
            return self.base_lr * 0.1 ** sum(1 for m in self.milestones if self.step_count >= m)
        return cosine_lr(self.base_lr, self.step_count, self.t_max)

    # -- moco.py:154-185 + optimizer_hook.py:25-50 ---------------------------
    def train_step(self, img_q, img_k, keys_all_ranks=None, taps=None):
        """One full step.  Returns dict(loss, acc1, acc5, logits, q, k, grads)."""
        out = self.forward_backward(img_q, img_k, keys_all_ranks=keys_all_ranks, taps=taps)
        self.apply_momentum(out['grads'])
        return out

    def train_step_accum(self, img_q, img_k, accum_steps):
        """passl/engine/loops/contrastive_learning_loop.py:31-88: the batch is cut into ``accum_steps``
        micro-batches; the MODEL is called once per micro-batch (so the key-encoder EMA, the
        micro-batch BatchNorm statistics and the enqueue all happen per micro-batch, the second
        micro-batch already sees the first one's keys in the queue), every loss is divided by
        accum_steps before backward, gradients add up, then ONE optimizer step and lr step.
        Returns dict(loss = sum of the scaled losses, grads = accumulated)."""
        N = img_q.shape[0]
        assert N % accum_steps == 0
        step = N // accum_steps
        total, loss = None, 0.0
        for i in range(accum_steps):
            sl = slice(i * step, (i + 1) * step)
            out = self.forward_backward(img_q[sl], img_k[sl], loss_scale=1.0 / accum_steps)
            loss = loss + out['loss'] / accum_steps
            if total is None:
                total = out['grads']
            else:
                for n in total:
                    total[n] = total[n] + out['grads'][n]
        self.apply_momentum(total)
        return dict(loss=loss, grads=total)

    def forward_backward(self, img_q, img_k, keys_all_ranks=None, taps=None, loss_scale=1.0):
        """Forward, EMA, enqueue and backward of (loss * loss_scale); no optimizer step."""
        tkeys = R.trainable_keys(self.q)
        for n in tkeys:
            self.q[n] = self.q[n].detach()
            self.q[n].requires_grad_(True)
            self.q[n].grad = None
        new_stats = {}
        q = R.encoder_forward(self.q, img_q, use_global_stats=False,
                              new_stats=new_stats, taps=taps, bf16=self.bf16)
        q = l2_normalize(q, axis=1)
        with torch.no_grad():
            # BN running stats are written by the q forward *before* the EMA
            # (paddle BN updates them in-place inside encoder_q(img_q), moco.py:158)
            for n, v in new_stats.items():
                self.q[n] = v
            self.momentum_update_key_encoder()
            k = R.encoder_forward(self.k, img_k, use_global_stats=True, bf16=self.bf16)
            k = l2_normalize(k, axis=1)
        l_pos = (q * k).sum(dim=1, keepdim=True)
        l_neg = q @ self.queue.clone().detach()
        loss, acc1, acc5, logits = contrastive_head(l_pos, l_neg, self.T)
        self.dequeue_and_enqueue(k if keys_all_ranks is None else keys_all_ranks)
        (loss * loss_scale).backward()
        grads = OrderedDict((n, self.q[n].grad.detach().clone()) for n in tkeys)
        return dict(loss=loss.detach(), acc1=acc1, acc5=acc5, logits=logits.detach(),
                    q=q.detach(), k=k, grads=grads)

    @torch.no_grad()
    def apply_momentum(self, grads):
        lr = self.lr()
        for n, g in grads.items():
            p = self.q[n].detach()
            g = g + self.wd * p
            v = self.velocity.get(n)
            v = g.clone() if v is None else self.mu * v + g
            self.velocity[n] = v
            self.q[n] = (p - lr * v).detach()
        self.step_count += 1     # LRSchedulerHook.train_iter_end
