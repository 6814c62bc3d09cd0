"""Oracle: one step of the v2 linear-probe recipes (tasks/ssl/simsiam/configs/simsiam_resnet50_lp_*.yaml and
tasks/ssl/mocov3/configs/mocov3_vit_base_patch16_224_lp_*.yaml): frozen encoder -> Linear classifier, CELoss, TopkAcc,
Momentum / MomentumLARC under TimmCosine, and the evaluation pass, torch-CPU fp32 / fp64.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Follows:

* passl/models/simsiam.py:128-147   SimSiamLinearProbe(ResNet): every parameter but ``fc.weight`` / ``fc.bias`` has
                                    stop_gradient = True, every BatchNorm ``_use_global_stats = True`` (running statistics
                                    also in train mode), fc ~ Normal(0, 0.01) / zero bias
* passl/models/mocov3.py:94-109     MoCoV3LinearProbe(MoCoV3ViT): the same for ``head.weight`` / ``head.bias``;
                                    forward = VisionTransformer.forward (vision_transformer.py:345-363):
                                    head(norm(blocks([cls | patches] + pos_embed))[:, 0])
* passl/loss/celoss.py:22-56, passl/loss/__init__.py:25-55   CombinedLoss([CELoss(weight 1.0)]): logits cast to fp32,
                                    mean cross entropy over hard labels, ``loss`` = weighted sum of the entries
* passl/metric/metrics.py:29-56     TopkAcc(topk=[1, 5]): paddle.metric.accuracy (FRACTION of rows whose label is among
                                    the k best scores) as python floats; ``metric`` = the first k
* passl/optimizer/momentum_larc.py:56-111   per parameter TENSOR: if |p| != 0 and |g| != 0:
                                    a = trust_coefficient |p| / (|g| + |p| wd + eps)  [clip: a = min(a / lr, 1)],
                                    g = a (g + wd p);   v = mu v + g;   p -= lr v     (otherwise the raw gradient, WITHOUT
                                    weight decay — the zero-initialised bias on its first step)
* passl/optimizer/momentum.py:73-160   g += wd p (wd != 0);  v = g on the first step, mu v + g after;  p -= lr v
* passl/optimizer/optimizer.py:117-125   the learning rate of a step is ``scheduler.get_lr()`` evaluated at the CURRENT
                                    ``last_epoch`` (not the cached ``last_lr``)
* passl/scheduler/lr_scheduler.py:22-77   TimmCosine (see ``timm_cosine``)
* passl/engine/loops/classification_loop.py:36-101   train_one_step: forward -> loss -> backward -> grad_sync ->
                                    optimizer.step -> clear_grad -> lr_step(global_step) for decay_unit 'step'
                                    (loop.py:224-225: lr_step(cur_epoch_id) after every epoch for 'epoch')
* passl/engine/loops/classification_loop.py:147-262  eval_one_dataset: model.eval(), per batch loss + metric, averaged
                                    with the batch sizes as weights

States: flat ``dict[str, Tensor]`` with the reference's state_dict names (v2 ResNet: ``conv1 / bn1 / layer*.* / fc``;
MoCoV3ViT: ``cls_token / pos_embed / patch_embed.proj / blocks.* / norm / head``), Linear weights [in, out].
Pinned by executing the reference's own sources — models, CombinedLoss, TopkAcc, Momentum, MomentumLARC, TimmCosine —
under the paddle shim: tests/golden/make_golden_linprobe_v2.py.
"""
import math
from collections import OrderedDict

import torch
import torch.nn.functional as F

from . import clas as C
from . import mocov3 as V
from . import resnet50 as R

HEAD = {'simsiam': ('fc.weight', 'fc.bias'), 'mocov3': ('head.weight', 'head.bias')}


def timm_cosine(last_epoch, learning_rate, T_max, warmup_steps=0, eta_min=0.0, warmup_start_lr=0.0,
                warmup_prefix=False):
    """TimmCosine.get_lr() at ``last_epoch`` (lr_scheduler.py:64-77)."""
    if last_epoch < warmup_steps:
        return float(max(0, last_epoch)) * (learning_rate - warmup_start_lr) / float(warmup_steps) + warmup_start_lr
    t_max = T_max
    if warmup_prefix:
        last_epoch = last_epoch - warmup_steps
        t_max = T_max - warmup_steps
    cur = last_epoch - (T_max * (last_epoch // T_max))
    return eta_min + 0.5 * (float(learning_rate) - eta_min) * (1 + math.cos(math.pi * cur / t_max))


def simsiam_state(gen, class_num=1000):
    """A 'pre-trained' trunk (calibrated running statistics, perturbed BatchNorm shifts: oracle/clas.py) + the probe's
    fresh classifier."""
    st = OrderedDict()
    for k, v in C.init_state(gen, num_classes=8).items():
        if k.startswith('backbone.'):
            st[k[len('backbone.'):]] = v
    st['fc.weight'] = torch.randn(2048, class_num, generator=gen) * 0.01
    st['fc.bias'] = torch.zeros(class_num)
    return st


def mocov3_state(gen, cfg, class_num=1000):
    st = OrderedDict()
    full = V.init_state(gen, **cfg)
    for k, v in full.items():
        if k.startswith('base_encoder.') and not k.startswith('base_encoder.head.'):
            st[k[len('base_encoder.'):]] = v
    # a pre-trained encoder has non-trivial LayerNorm affines and biases
    for k in list(st):
        if k.endswith('.bias'):
            st[k] = torch.randn(st[k].shape, generator=gen) * 0.02
        elif '.norm' in k and k.endswith('.weight') or k == 'norm.weight':
            st[k] = 1.0 + torch.randn(st[k].shape, generator=gen) * 0.05
    st['head.weight'] = torch.randn(cfg['embed_dim'], class_num, generator=gen) * 0.01
    st['head.bias'] = torch.zeros(class_num)
    return st


def features(kind, st, x, cfg=None):
    """The frozen encoder (no gradient reaches it: every parameter has stop_gradient)."""
    with torch.no_grad():
        if kind == 'simsiam':
            trunk = {'0.' + k: v for k, v in st.items() if not k.startswith('fc.')}
            return R.trunk_forward(trunk, x, True, None, maxpool=True).mean(dim=(2, 3))
        enc = {'base_encoder.' + k: v for k, v in st.items() if not k.startswith('head.')}
        return V.vit_features(enc, x, cfg)


def scores(kind, st, x, cfg=None):
    w, b = HEAD[kind]
    return features(kind, st, x, cfg) @ st[w] + st[b]


def topk_acc(s, labels, topk=(1, 5)):
    """Fractions in [0, 1] (paddle.metric.accuracy), ties to the lower index."""
    return [a / 100.0 for a in C.accuracy(s, labels, topk)]


class LinearProbeOracle:
    def __init__(self, kind, class_num=1000, seed=0, cfg=None, optimizer='MomentumLARC', lr=0.1, momentum=0.9,
                 weight_decay=0.0, trust_coefficient=0.001, clip=False, eps=1e-8, dtype=torch.float32):
        assert kind in HEAD and optimizer in ('Momentum', 'MomentumLARC')
        gen = torch.Generator().manual_seed(seed)
        self.kind, self.cfg = kind, cfg
        st = simsiam_state(gen, class_num) if kind == 'simsiam' else mocov3_state(gen, cfg, class_num)
        self.st = OrderedDict((k, v.to(dtype)) for k, v in st.items())
        self.optimizer, self.lr_value, self.mu, self.wd = optimizer, lr, momentum, weight_decay
        self.tc, self.clip, self.eps = trust_coefficient, clip, eps
        self.exp_avg = {}
        self.step_count = 0

    def lr(self):
        return self.lr_value(self.step_count) if callable(self.lr_value) else self.lr_value

    def update(self, grads):
        lr = self.lr()
        for n, g in grads.items():
            p = self.st[n].detach()
            if self.optimizer == 'MomentumLARC':
                pn, gn = p.norm(), g.norm()
                if pn != 0 and gn != 0:
                    a = self.tc * pn / (gn + pn * self.wd + self.eps)
                    if self.clip:
                        a = torch.clamp(a / lr, max=1.0)
                    g = a * (g + self.wd * p)
                v = self.exp_avg.get(n, torch.zeros_like(p)) * self.mu + g
            else:
                if self.wd != 0.0:
                    g = g + self.wd * p
                v = g.clone() if n not in self.exp_avg else self.exp_avg[n] * self.mu + g
            self.exp_avg[n] = v
            self.st[n] = p - lr * v
        self.step_count += 1

    def train_step(self, x, labels):
        w, b = HEAD[self.kind]
        for n in (w, b):
            self.st[n] = self.st[n].detach().requires_grad_(True)
        s = scores(self.kind, self.st, x, self.cfg)
        loss = F.cross_entropy(s, labels)
        loss.backward()
        grads = OrderedDict((n, self.st[n].grad.detach().clone()) for n in (w, b))
        lr = self.lr()
        self.update(grads)
        top1, top5 = topk_acc(s.detach(), labels)
        return dict(loss=loss.detach(), scores=s.detach(), grads=grads, top1=top1, top5=top5, lr=lr)

    @torch.no_grad()
    def evaluate(self, batches):
        """eval_one_dataset (classification_loop.py:147-262), one rank: averages weighted by the batch sizes."""
        tot = dict(CELoss=0.0, loss=0.0, top1=0.0, top5=0.0)
        n_all = 0
        for x, labels in batches:
            s = scores(self.kind, self.st, x, self.cfg)
            loss = float(F.cross_entropy(s, labels))
            top1, top5 = (float(a) for a in topk_acc(s, labels))
            n = x.shape[0]
            for k, v in (('CELoss', loss), ('loss', loss), ('top1', top1), ('top5', top5)):
                tot[k] += v * n
            n_all += n
        out = {k: v / n_all for k, v in tot.items()}
        out['metric'] = out['top1']
        return out
