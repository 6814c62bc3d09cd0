"""Oracle: one SimSiam pre-training step (ResNet-50 encoder with a 3-layer BatchNorm projector in place of its fc,
2-layer predictor, symmetric negative cosine similarity with stop-gradient, momentum SGD with two parameter groups),
torch-CPU fp32 / fp64.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Follows, line by line:

* passl/models/simsiam.py:36-69     SimSiamPretain: ``encoder = ResNet(BottleneckBlock, 50, class_num=dim,
                                    zero_init_residual=True)``; ``encoder.fc`` := Linear(no bias)-BN1D-ReLU,
                                    Linear(no bias)-BN1D-ReLU, the original fc (Linear 2048 -> dim WITH a bias whose
                                    gradient is switched off), BN1D(no gamma / beta); predictor := Linear(no bias)-BN1D-
                                    ReLU-Linear(bias)
* passl/models/simsiam.py:71-95     forward: z = encoder(x) and p = predictor(z) for each view SEPARATELY (BatchNorm
                                    statistics per view), loss = -(cos(p1, z2.detach()).mean() + cos(p2, z1.detach()).mean()) / 2
* passl/models/resnet.py:52-73      v2 ResNet = paddle.vision.models.resnet.ResNet (Paddle wheel, 2.4 line; the tree's own
                                    copy is passl_v110/modeling/backbones/resnetimagenet.py:111-253: stem 7x7/2 + BN + ReLU
                                    + maxpool 3x3/2, bottlenecks 3/4/6/3 with the stride on conv2, avgpool, flatten, fc) with
                                    the last BatchNorm of every residual branch zero-initialised
* passl/optimizer/momentum.py:25-158  g += wd * p;  v = g (first step) | mu * v + g;  p -= lr * v  — every parameter that
                                    has a gradient, BatchNorm affine and biases included
* passl/optimizer/__init__.py:68-122,193-212 + tasks/ssl/simsiam/configs/simsiam_resnet50_pt_in1k_1n8c_dp_fp32.yaml
                                    parameter groups by name: ``encoder`` follows the schedule (TimmCosine per epoch),
                                    ``predictor`` keeps lr = 0.1 (SimSiam's fixed predictor rate)

The trunk is oracle/resnet50.py's (pinned by the reference's v110 sources); state = flat ``dict[str, Tensor]`` with the
reference's state_dict names (``encoder.*``, ``predictor.*``), Linear weights [in, out].

[Paddle-semantics] assumptions: nn.CosineSimilarity(axis=1) = sum(a*b) / max(|a|*|b|, 1e-8); BatchNorm1D as BatchNorm2D
(momentum 0.9, eps 1e-5, biased running variance); data parallel runs convert every BatchNorm to SyncBatchNorm
(simsiam.py:160-162) — world size 1 here.
"""
import math
from collections import OrderedDict

import torch
import torch.nn.functional as F

from . import resnet50 as R

FROZEN = ('encoder.fc.6.bias',)


def init_state(gen, dim=2048, pred_dim=512, zero_init_residual=True):
    st = OrderedDict()
    trunk = R.init_encoder_state(gen, neck='LinearNeck', out_channels=8)        # '0.<key>' trunk entries
    for k, v in trunk.items():
        if k.startswith('0.'):
            st['encoder.' + k[2:]] = v
    if zero_init_residual:
        for k in list(st):
            if k.endswith('.bn3.weight'):
                st[k] = torch.zeros_like(st[k])
    prev = 2048

    def lin(name, cin, cout, bias):
        bound = 1.0 / math.sqrt(cin)
        st[name + '.weight'] = (torch.rand(cin, cout, generator=gen) * 2 - 1) * bound
        if bias:
            st[name + '.bias'] = torch.zeros(cout)

    def bn(name, c, affine=True):
        if affine:
            st[name + '.weight'], st[name + '.bias'] = torch.ones(c), torch.zeros(c)
        st[name + '._mean'], st[name + '._variance'] = torch.zeros(c), torch.ones(c)
    lin('encoder.fc.0', prev, prev, False); bn('encoder.fc.1', prev)
    lin('encoder.fc.3', prev, prev, False); bn('encoder.fc.4', prev)
    lin('encoder.fc.6', prev, dim, True); bn('encoder.fc.7', dim, affine=False)
    lin('predictor.0', dim, pred_dim, False); bn('predictor.1', pred_dim)
    lin('predictor.3', pred_dim, dim, True)
    return st


def is_stat(k):
    return k.endswith('._mean') or k.endswith('._variance')


def trainable_keys(st):
    return [k for k in st if not is_stat(k) and k not in FROZEN]


def group_of(k):
    """passl/optimizer/__init__.py:97-105: re.match(group name, parameter name)."""
    return 'predictor' if k.startswith('predictor') else 'encoder'


def bn1d(st, key, x, new_stats):
    mean = x.mean(dim=0)
    var = x.var(dim=0, unbiased=False)
    with torch.no_grad():
        new_stats[key + '._mean'] = R.BN_MOMENTUM * st[key + '._mean'] + (1 - R.BN_MOMENTUM) * mean.detach()
        new_stats[key + '._variance'] = R.BN_MOMENTUM * st[key + '._variance'] + (1 - R.BN_MOMENTUM) * var.detach()
    inv = torch.rsqrt(var + R.BN_EPS)
    if key + '.weight' in st:
        return (x - mean[None]) * (inv * st[key + '.weight'])[None] + st[key + '.bias'][None]
    return (x - mean[None]) * inv[None]


def encode(st, x):
    """One view through encoder + predictor; the running statistics in ``st`` are advanced (each view is its own
    BatchNorm batch, so a step advances them twice)."""
    trunk = {'0.' + k[len('encoder.'):]: v for k, v in st.items()
             if k.startswith('encoder.') and not k.startswith('encoder.fc.')}
    new_stats = {}
    f = R.trunk_forward(trunk, x, False, new_stats, maxpool=True)
    for k, v in new_stats.items():
        st['encoder.' + k[2:]] = v
    f = F.adaptive_avg_pool2d(f, 1).flatten(1)
    ns = {}
    h = F.relu(bn1d(st, 'encoder.fc.1', f @ st['encoder.fc.0.weight'], ns))
    h = F.relu(bn1d(st, 'encoder.fc.4', h @ st['encoder.fc.3.weight'], ns))
    z = bn1d(st, 'encoder.fc.7', h @ st['encoder.fc.6.weight'] + st['encoder.fc.6.bias'], ns)
    h = F.relu(bn1d(st, 'predictor.1', z @ st['predictor.0.weight'], ns))
    p = h @ st['predictor.3.weight'] + st['predictor.3.bias']
    st.update(ns)
    return z, p


def cosine(a, b, eps=1e-8):
    w12 = (a * b).sum(1)
    return w12 / ((a * a).sum(1) * (b * b).sum(1)).sqrt().clamp_min(eps)


class SimSiamOracle:
    def __init__(self, seed=0, dim=2048, pred_dim=512, lr=0.1, predictor_lr=0.1, momentum=0.9, weight_decay=1e-4,
                 zero_init_residual=True, dtype=torch.float32):
        gen = torch.Generator().manual_seed(seed)
        self.st = OrderedDict((k, v.to(dtype)) for k, v in init_state(gen, dim, pred_dim, zero_init_residual).items())
        self.lr_value, self.predictor_lr = lr, predictor_lr
        self.mu, self.wd = momentum, weight_decay
        self.velocity = {}
        self.step_count = 0

    def lr(self):
        return self.lr_value(self.step_count) if callable(self.lr_value) else self.lr_value

    def forward_backward(self, x1, x2):
        tk = trainable_keys(self.st)
        for n in tk:
            self.st[n] = self.st[n].detach().requires_grad_(True)
        z1, p1 = encode(self.st, x1)
        z2, p2 = encode(self.st, x2)
        loss = -(cosine(p1, z2.detach()).mean() + cosine(p2, z1.detach()).mean()) * 0.5
        loss.backward()
        grads = OrderedDict((n, self.st[n].grad.detach().clone()) for n in tk)
        return dict(loss=loss.detach(), z1=z1.detach(), z2=z2.detach(), p1=p1.detach(), p2=p2.detach(), grads=grads)

    def apply_momentum(self, grads):
        lr_e, lr_p = self.lr(), self.predictor_lr
        for n, g in grads.items():
            p = self.st[n].detach()
            g = g + self.wd * p
            v = self.velocity.get(n)
            v = g.clone() if v is None else self.mu * v + g
            self.velocity[n] = v
            self.st[n] = (p - (lr_p if group_of(n) == 'predictor' else lr_e) * v).detach()
        self.step_count += 1

    def train_step(self, x1, x2):
        out = self.forward_backward(x1, x2)
        self.apply_momentum(out['grads'])
        return out
