"""Oracle: one linear-probe step (frozen ResNet-50 trunk + ClasHead), torch-CPU fp32/fp64.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Follows:

* passl_v110/modeling/architectures/clas.py:25-77  Classification: train_iter = backbone -> head ->
  head.loss(outs, label); test_iter = the class scores under no_grad
* passl_v110/modeling/backbones/resnet.py:90-106  _freeze_stages: frozen_stages = 4 (configs/moco/
  moco_clas_r50.yaml) freezes conv1/bn1 and layer1-4: parameters not trainable, every BatchNorm uses
  its running statistics (modules/freeze.py)
* passl_v110/modeling/heads/clas_head.py:22-72  ClasHead: AdaptiveAvgPool2D(1) -> reshape ->
  Linear(2048, 1000) (Normal(0, 0.01) / zero bias); loss = CrossEntropyLoss; acc1 / acc5 = percentage
  of rows whose label is among the top-1 / top-5 scores
* configs/moco/moco_clas_r50.yaml: Momentum(lr 30.0, momentum 0.9 (Paddle default), weight_decay 0.0)
  over the trainable parameters (the head), MultiStepDecay(milestones [60, 80] epochs, gamma 0.1)

[Paddle-semantics] assumptions: as oracle/resnet50.py and oracle/moco.py (BatchNorm with
use_global_stats, Linear [in,out], momentum rule v = mu v + g; p -= lr v); topk ties resolve to the
lower index.
"""
import math
from collections import OrderedDict

import torch
import torch.nn.functional as F

from .resnet50 import init_encoder_state, trunk_forward


def init_state(gen, num_classes=1000, width_div=1):
    """backbone.* = the ResNet keys, head.fc_cls.{weight [in,out], bias}.  The BatchNorm running
    statistics are CALIBRATED (batch statistics of one seed-defined 8 x 64 x 64 batch, as a pre-trained
    checkpoint would carry) so that the frozen trunk produces O(1) features; BN shifts are perturbed
    so that the frozen BN is not a pure normalisation."""
    enc = init_encoder_state(gen, width_div=width_div)
    bb = OrderedDict((k, v) for k, v in enc.items() if k.startswith('0.'))
    for k in bb:
        if k.endswith('.bias'):
            bb[k] = torch.randn(bb[k].shape, generator=gen) * 0.1
    new = {}
    with torch.no_grad():
        trunk_forward(bb, torch.randn(8, 3, 64, 64, generator=gen), use_global_stats=False, new_stats=new)
    for k, v in new.items():          # new = 0.9 * running + 0.1 * batch with running = (0, 1)
        bb[k] = v / 0.1 if k.endswith('._mean') else (v - 0.9) / 0.1
    st = OrderedDict(('backbone.' + k[2:], v) for k, v in bb.items())
    cin = 2048 // width_div
    st['head.fc_cls.weight'] = torch.randn(cin, num_classes, generator=gen) * 0.01
    st['head.fc_cls.bias'] = torch.zeros(num_classes)
    return st


def backbone_state(st):
    return OrderedDict(('0.' + k[len('backbone.'):], v) for k, v in st.items() if k.startswith('backbone.'))


def accuracy(scores, labels, topk=(1, 5)):
    lab = scores.gather(1, labels.view(-1, 1))
    idx = torch.arange(scores.shape[1]).view(1, -1)
    rank = ((scores > lab) | ((scores == lab) & (idx < labels.view(-1, 1)))).sum(dim=1)
    return [(rank < k).double().sum() * 100.0 / scores.shape[0] for k in topk]


def frozen_keys(st, frozen_stages):
    """Parameters _freeze_stages (resnet.py:90-106) makes non-trainable: conv1/bn1 for frozen_stages >= 0
    plus layer1..layer<frozen_stages>; BatchNorm running statistics are never trainable."""
    out = set()
    for k in st:
        if k.endswith('._mean') or k.endswith('._variance'):
            out.add(k)
        elif k.startswith('backbone.') and frozen_stages >= 0:
            name = k[len('backbone.'):]
            stage = int(name[5]) if name.startswith('layer') else 0
            if stage <= frozen_stages:
                out.add(k)
    return out


def clas_forward(st, img, labels, frozen_stages=4, new_stats=None):
    if frozen_stages >= 4:
        with torch.no_grad():
            feat = trunk_forward(backbone_state(st), img, use_global_stats=True)     # [N, 2048, h, w]
    else:
        feat = trunk_forward(backbone_state(st), img, use_global_stats=False, new_stats=new_stats,
                             frozen_stages=frozen_stages)
    x = feat.mean(dim=(2, 3))
    scores = x @ st['head.fc_cls.weight'] + st['head.fc_cls.bias']
    loss = F.cross_entropy(scores, labels)
    acc1, acc5 = accuracy(scores.detach(), labels)
    return dict(loss=loss, acc1=acc1, acc5=acc5, scores=scores, feat=x)


def multistep_lr(base_lr, epoch, milestones, gamma=0.1):
    return base_lr * gamma ** sum(1 for m in milestones if epoch >= m)


class ClasOracle:
    def __init__(self, num_classes=1000, seed=0, lr=30.0, momentum=0.9, dtype=torch.float32, width_div=1,
                 frozen_stages=4):
        self.frozen_stages = frozen_stages
        gen = torch.Generator().manual_seed(seed)
        self.st = OrderedDict((k, v.to(dtype)) for k, v in init_state(gen, num_classes, width_div).items())
        self.lr_value, self.mu = lr, momentum
        self.vel = OrderedDict()
        self.step_count = 0

    def lr(self):
        return self.lr_value(self.step_count) if callable(self.lr_value) else self.lr_value

    def train_step(self, img, labels):
        fz = frozen_keys(self.st, self.frozen_stages)
        tk = [n for n in self.st if n not in fz]
        for n in tk:
            self.st[n] = self.st[n].detach().requires_grad_(True)
        new_stats = {}
        out = clas_forward(self.st, img, labels, self.frozen_stages, new_stats)
        out['loss'].backward()
        with torch.no_grad():
            for k, v in new_stats.items():        # running statistics of the BatchNorms still training
                self.st['backbone.' + k[2:]] = v
        grads = OrderedDict((n, self.st[n].grad.detach().clone()) for n in tk)
        lr = self.lr()
        with torch.no_grad():
            for n, g in grads.items():
                v = self.mu * self.vel.get(n, torch.zeros_like(g)) + g
                self.vel[n] = v
                self.st[n] = (self.st[n].detach() - lr * v)
        self.step_count += 1
        res = {k: (v.detach() if torch.is_tensor(v) else v) for k, v in out.items()}
        res['grads'] = grads
        return res
