"""Oracle: one SimCLR training step, torch-CPU fp32 (+ numpy fp64 head).

TEST INFRASTRUCTURE (see oracle/__init__.py).  Follows:

* passl_v110/modeling/architectures/simclr.py:52-61   train_iter: concat the two views, ONE
  encoder pass over 2N images, l2_normalize, split, head
* passl_v110/modeling/backbones/resnetsimclr.py:25-91 over resnetcifar.py:216-333: the R50
  topology of resnetimagenet.py WITHOUT the stem max-pool (resnetcifar.py:275 is commented out),
  avgpool inside the backbone (with_pool=True), convs initialised XavierNormal(fan_in=None,
  fan_out=0) (resnetcifar.py:62-70 ...), `init_parameters()` commented out (resnetsimclr.py:63)
* passl_v110/modeling/necks/base_neck.py:209-239  NonLinearNeckfc3: fc-BN1D-ReLU-fc-BN1D-ReLU-
  fc-BN1D then l2_normalize; Linear weights ~ N(0, 0.01) (modules/init.py:406-412)
* passl_v110/modeling/heads/simclr_contrastive_head.py:42-102  NT-Xent over [ab|aa] / [ba|bb]
  with LARGE_NUM self-masks + 3 x CO2 (two KL terms), acc1 = top-1 of logits_ab (a FRACTION)
* passl_v110/hooks/optimizer_hook.py:25-50  LARS branch: clear_gradients / backward / minimize
* passl_v110/solver/builder.py:41-66 + lr_scheduler.py:105-139  simclrCosineWarmup

[Paddle-semantics] assumptions (Paddle is not in the reference tree; see oracle/README.md):
* fluid.layers.l2_normalize(x, axis) = x / sqrt(sum(x^2) + 1e-12)
* softmax_with_cross_entropy(soft_label=True) = -sum(label * log_softmax(logits)) per row [N,1]
* kl_div(input, label, 'batchmean') = sum(label*(log(label) - input) where label>0)/N and the
  op has NO gradient w.r.t. `label` (kldiv_loss_grad only produces x_grad) — the CO2 term
  therefore back-propagates only through log_a / log_b
* XavierNormal(fan_in=None, fan_out=0): fan_out=0 is "not None" -> std = sqrt(2/(fan_in+0))
  with fan_in = Cin*kh*kw
* nn.Linear default bias initialiser = 0; BatchNorm1D = BatchNorm2D semantics on [N,C]
* LarsMomentumOptimizer (lars_momentum op): for every parameter
      local_lr = lr                                  if wd == 0 or |p| == 0 or |g| == 0
               = lr*coeff*|p| / (|g| + wd*|p| + eps) otherwise
      v = mu*v + local_lr*(g + wd*p);  p = p - v
  `exclude_from_weight_decay` is matched against the parameter's Paddle NAME
  ('conv2d_0.w_0', 'batch_norm2d_0.w_0', 'linear_0.b_0' ...): the yaml's strings
  ["scale","offset",".bias"] match no dygraph auto-name, so NO parameter is excluded.
* LinearWarmup/LRScheduler: lr(t) = end_lr*t/warmup for t < warmup, then the wrapped scheduler
  stepped to epoch t - warmup;  Cosinesimclr: lr*(1+cos(pi*t/T_max))/2.
"""
import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

from . import resnet50 as R
from .bf16 import round_act, round_weight, round_grad

LARGE_NUM = 1e9


# ------------------------------------------------------------------ state / init
def neck_keys():
    ks = []
    for i, kind in enumerate(['fc', 'bn', None, 'fc', 'bn', None, 'fc', 'bn']):
        if kind == 'fc':
            ks += ['1.mlp.%d.weight' % i, '1.mlp.%d.bias' % i]
        elif kind == 'bn':
            ks += ['1.mlp.%d%s' % (i, s) for s in ('.weight', '.bias', '._mean', '._variance')]
    return ks


def basic_conv_specs(layers=(2, 2, 2, 2)):
    """[(name, cout, cin, k, stride, pad, bn)] of the BasicBlock trunk (depth 18: 2-2-2-2; 34: 3-4-6-3) in
    construction order — resnetcifar.py:41-118 (block) and :283-311 (_make_layer: a downsample conv + BN on the
    first block of a stage whose stride or width changes)."""
    specs = [('conv1', 64, 3, 7, 2, 3, 'bn1')]
    inplanes = 64
    for li, (planes, blocks) in enumerate(zip((64, 128, 256, 512), layers), start=1):
        stride = 1 if li == 1 else 2
        for b in range(blocks):
            p = 'layer%d.%d' % (li, b)
            s_ = stride if b == 0 else 1
            specs.append((p + '.conv1', planes, inplanes, 3, s_, 1, p + '.bn1'))
            specs.append((p + '.conv2', planes, planes, 3, 1, 1, p + '.bn2'))
            if b == 0 and (s_ != 1 or inplanes != planes):
                specs.append((p + '.downsample.0', planes, inplanes, 1, s_, 0, p + '.downsample.1'))
            inplanes = planes
    return specs


def init_encoder_state(gen, in_channels=2048, hid_channels=2048, out_channels=128, depth=50):
    """Keys '0.<backbone>' / '1.<neck>' of nn.Sequential(ResNetsimclr, NonLinearNeckfc3).  depth 18 / 34: the
    BasicBlock trunk (configs/simclr/simclr_r18_cifar10.yaml: in / hid channels 512)."""
    st = OrderedDict()
    specs = R.conv_specs() if depth == 50 else basic_conv_specs({18: (2, 2, 2, 2), 34: (3, 4, 6, 3)}[depth])
    for name, cout, cin, k, _s, _p, bn in specs:
        std = math.sqrt(2.0 / (cin * k * k))        # XavierNormal(fan_in=None, fan_out=0)
        st['0.' + name + '.weight'] = torch.randn(cout, cin, k, k, generator=gen) * std
        st['0.' + bn + '.weight'] = torch.ones(cout)
        st['0.' + bn + '.bias'] = torch.zeros(cout)
        st['0.' + bn + '._mean'] = torch.zeros(cout)
        st['0.' + bn + '._variance'] = torch.ones(cout)
    dims = [(in_channels, hid_channels), (hid_channels, hid_channels), (hid_channels, out_channels)]
    for (i_fc, i_bn), (cin, cout) in zip(((0, 1), (3, 4), (6, 7)), dims):
        st['1.mlp.%d.weight' % i_fc] = torch.randn(cin, cout, generator=gen) * 0.01   # [in,out]
        st['1.mlp.%d.bias' % i_fc] = torch.zeros(cout)
        st['1.mlp.%d.weight' % i_bn] = torch.ones(cout)
        st['1.mlp.%d.bias' % i_bn] = torch.zeros(cout)
        st['1.mlp.%d._mean' % i_bn] = torch.zeros(cout)
        st['1.mlp.%d._variance' % i_bn] = torch.ones(cout)
    return st


def fluid_l2_normalize(x, axis=-1, eps=1e-12):
    return x * torch.rsqrt(x.pow(2).sum(dim=axis, keepdim=True) + eps)


def encoder_forward(st, x, new_stats=None, taps=None, bf16=False, rec=None):
    """ResNetsimclr(depth=50, with_pool=True) + NonLinearNeckfc3(with_avg_pool=False), train-mode
    BN everywhere (SimCLR has a single encoder).
    ``bf16``: bf16-emulating mode (oracle/bf16.py, resnet50.trunk_forward): the trunk's storage points,
    then the product's projector contract — pooled features and the inputs of the 2nd / 3rd Linear are
    stored in bf16 (value and gradient rounded), Linear operands are the bf16 weight copies, Linear
    outputs / BatchNorm1D / ReLU / normalisation are fp32, and the gradient entering a Linear's backward
    is rounded (its GEMM operands are bf16)."""
    x = R.trunk_forward(st, x, False, new_stats, taps, maxpool=False, bf16=bf16, rec=rec)
    x = F.adaptive_avg_pool2d(x, 1).reshape(x.shape[0], -1)      # backbone avgpool + squeeze

    def keep(key, t):
        # per-layer teacher forcing (tests/test_layers_gpu.py): stored tensor + its gradient
        if rec is not None:
            t.retain_grad()
            rec[key] = t
        return t

    for i_fc, i_bn, relu in ((0, 1, True), (3, 4, True), (6, 7, False)):
        if bf16:
            x = keep('mlp.%d.x' % i_fc, round_act(x))
            x = round_grad(R._matmul(x, round_weight(st['1.mlp.%d.weight' % i_fc]))) + st['1.mlp.%d.bias' % i_fc]
        else:
            x = x @ st['1.mlp.%d.weight' % i_fc] + st['1.mlp.%d.bias' % i_fc]
        x = keep('mlp.%d.y' % i_fc, x)
        x = R.batch_norm(x, st, '1.mlp.%d' % i_bn, False, new_stats)
        if relu:
            x = F.relu(x)
        x = keep('mlp.%d.z' % i_bn, x)
    return fluid_l2_normalize(x, -1)


# ------------------------------------------------------------------ head
def simclr_head(h1, h2, temperature, co2_weight=3.0):
    """SimCLRContrastiveHead.forward (simclr_contrastive_head.py:42-102).  Returns loss, acc1
    (fraction), and the four logit matrices."""
    B = h1.shape[0]
    dt = h1.dtype
    masks = torch.eye(B, dtype=dt)
    labels = torch.cat([torch.eye(B, dtype=dt), torch.zeros(B, B, dtype=dt)], dim=1)
    aa = h1 @ h1.t() / temperature - masks * LARGE_NUM
    bb = h2 @ h2.t() / temperature - masks * LARGE_NUM
    ab = h1 @ h2.t() / temperature
    ba = h2 @ h1.t() / temperature
    loss_a = -(labels * torch.log_softmax(torch.cat([ab, aa], 1), dim=1)).sum(1, keepdim=True)
    loss_b = -(labels * torch.log_softmax(torch.cat([ba, bb], 1), dim=1)).sum(1, keepdim=True)
    contrast = loss_a + loss_b
    logit_a = torch.cat([aa, ab - masks * LARGE_NUM], 1)
    logit_b = torch.cat([ba - masks * LARGE_NUM, bb], 1)
    log_a, log_b = torch.log_softmax(logit_a, 1), torch.log_softmax(logit_b, 1)
    a, b = torch.softmax(logit_a, 1).detach(), torch.softmax(logit_b, 1).detach()  # no grad to label
    kl_1 = F.kl_div(log_a, b, reduction='batchmean')
    kl_2 = F.kl_div(log_b, a, reduction='batchmean')
    loss = (contrast + co2_weight * (kl_1 + kl_2)).mean()
    with torch.no_grad():
        acc1 = (ab.argmax(dim=1) == torch.arange(B)).float().mean()
    return loss, acc1, dict(aa=aa, ab=ab, ba=ba, bb=bb)


def simclr_head_f64(h1, h2, temperature, co2_weight=3.0):
    """Same head in numpy float64 with explicit masking (no 1e9 arithmetic): the spot check."""
    a = np.asarray(h1, dtype=np.float64)
    b = np.asarray(h2, dtype=np.float64)
    B = a.shape[0]
    aa, ab, ba, bb = a @ a.T / temperature, a @ b.T / temperature, b @ a.T / temperature, b @ b.T / temperature
    eye = np.eye(B, dtype=bool)
    ninf = -np.inf

    def lse(x):
        m = x.max(axis=1, keepdims=True)
        return m[:, 0] + np.log(np.exp(x - m).sum(axis=1))

    aam, bbm = np.where(eye, ninf, aa), np.where(eye, ninf, bb)
    abm, bam = np.where(eye, ninf, ab), np.where(eye, ninf, ba)
    pos = np.diag(ab)
    ce = (lse(np.concatenate([ab, aam], 1)) - pos) + (lse(np.concatenate([ba, bbm], 1)) - pos)
    x, y = np.concatenate([aam, abm], 1), np.concatenate([bam, bbm], 1)
    lx, ly = lse(x), lse(y)
    pa, pb = np.exp(x - lx[:, None]), np.exp(y - ly[:, None])
    with np.errstate(invalid='ignore'):
        d = np.where(np.isfinite(x), y - x, 0.0)
    kl = ((pb - pa) * d).sum() / B                       # kl_1 + kl_2 (symmetrised KL)
    loss = ce.mean() + co2_weight * kl
    acc1 = float((ab.argmax(axis=1) == np.arange(B)).mean())
    return float(loss), acc1


# ------------------------------------------------------------------ solver
def simclr_lr(t, lr, warmup_steps, t_max):
    """simclrCosineWarmup value at scheduler epoch t (start_lr 0, end_lr lr)."""
    if t < warmup_steps:
        return lr * float(t) / float(warmup_steps)
    return lr * (1 + math.cos(math.pi * (t - warmup_steps) / t_max)) / 2


def paddle_param_names(keys):
    """Dygraph auto-names in construction order for the exclude_from_weight_decay substring test."""
    counters, names = {}, {}
    for k in keys:
        if k.endswith('._mean') or k.endswith('._variance'):
            continue
        mod, leaf = k.rsplit('.', 1)
        kind = 'batch_norm2d' if ('bn' in mod.split('.')[-1] or mod.endswith('downsample.1')) else \
            ('conv2d' if 'conv' in mod or 'downsample.0' in mod else None)
        if mod.startswith('1.mlp.'):
            kind = 'linear' if int(mod.split('.')[-1]) in (0, 3, 6) else 'batch_norm1d'
        tag = (kind, mod)
        if tag not in names:
            idx = counters.get(kind, 0)
            counters[kind] = idx + 1
            names[tag] = '%s_%d' % (kind, idx)
        names[k] = names[tag] + ('.w_0' if leaf == 'weight' else '.b_0')
    return {k: v for k, v in names.items() if isinstance(k, str)}


class SimCLROracle:
    def __init__(self, T=0.1, lr=64.0, warmup_steps=3127, t_max=28152, momentum=0.9,
                 lars_coeff=0.001, lars_weight_decay=1e-4, epsilon=0.0,
                 exclude=('scale', 'offset', '.bias'), seed=0, bf16=False, depth=50, in_channels=2048,
                 hid_channels=2048):
        gen = torch.Generator().manual_seed(seed)
        self.T = T
        # bf16=True: bf16-emulating encoder (encoder_forward); head, LARS and lr stay fp32 as in the product
        self.bf16 = bf16
        self.lr0, self.warmup_steps, self.t_max = lr, warmup_steps, t_max
        self.mu, self.coeff, self.wd, self.eps = momentum, lars_coeff, lars_weight_decay, epsilon
        # depth 18 / 34: state, LARS and schedule only (train_step restates the bottleneck trunk; the R18 goldens are
        # the reference's own forward / backward, tests/golden/make_golden_simclr_r18.py)
        self.depth = depth
        self.st = init_encoder_state(gen, in_channels, hid_channels, depth=depth)
        pnames = paddle_param_names(list(self.st.keys()))
        self.excluded = {k for k, n in pnames.items() if any(e in n for e in exclude)}
        self.velocity = OrderedDict()
        self.step_count = 0

    def lr(self):
        return simclr_lr(self.step_count, self.lr0, self.warmup_steps, self.t_max)

    def train_step(self, img_q, img_k, taps=None):
        assert self.depth == 50, 'the restated trunk is the bottleneck one'
        tkeys = R.trainable_keys(self.st)
        for n in tkeys:
            self.st[n] = self.st[n].detach().requires_grad_(True)
        new_stats = {}
        con = encoder_forward(self.st, torch.cat([img_q, img_k]), new_stats, taps, bf16=self.bf16)
        con = fluid_l2_normalize(con, -1)                    # simclr.py:57 (second normalisation)
        q, k = con[:img_q.shape[0]], con[img_q.shape[0]:]
        loss, acc1, mats = simclr_head(q, k, self.T)
        loss.backward()
        grads = OrderedDict((n, self.st[n].grad.detach().clone()) for n in tkeys)
        with torch.no_grad():
            for n, v in new_stats.items():
                self.st[n] = v
        self.apply_lars(grads)
        return dict(loss=loss.detach(), acc1=acc1, q=q.detach(), k=k.detach(), grads=grads,
                    mats={n: m.detach() for n, m in mats.items()})

    @torch.no_grad()
    def apply_lars(self, grads):
        lr = self.lr()
        for n, g in grads.items():
            p = self.st[n].detach()
            wd = 0.0 if n in self.excluded else self.wd
            pn, gn = float(p.double().norm()), float(g.double().norm())
            local_lr = lr
            if wd > 0 and pn > 0 and gn > 0:
                local_lr = lr * self.coeff * pn / (gn + wd * pn + self.eps)
            v = self.velocity.get(n)
            v = torch.zeros_like(p) if v is None else v
            v = self.mu * v + local_lr * (g + wd * p)
            self.velocity[n] = v
            self.st[n] = (p - v).detach()
        self.step_count += 1
