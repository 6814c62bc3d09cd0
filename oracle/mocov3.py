"""Oracle: one MoCo-v3 pre-training step (ViT base encoder + projector + predictor, momentum encoder, cross-rank
symmetric InfoNCE, AdamW), torch-CPU fp32/fp64.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Follows, line by line:

* passl/models/mocov3.py:36-91    MoCoV3ViT = VisionTransformer + FIXED 2-D sin-cos position embedding
                                  (build_2d_sincos_position_embedding: [sin w, cos w, sin h, cos h], the class
                                  position is zero), qkv / Linear / patch-embed init, ``stop_grad_conv1``
* passl/models/vision_transformer.py:84-249, 256-364   Mlp, Attention, Block (pre-norm), PatchEmbed (16x16/s16 conv
                                  with bias), forward_features: [cls | patches] + pos_embed -> blocks -> norm -> x[:, 0]
* passl/models/mocov3.py:111-166  MoCoV3Pretrain: ``base_encoder.head`` is replaced by the 3-layer projector
                                  (Linear(no bias)-BN1D-ReLU x2, Linear(no bias)-BN1D(no affine)), predictor = 2 layers
                                  of the same kind; momentum encoder = CosineEMA(Sequential(base_encoder, predictor))
* passl/models/mocov3.py:185-222  forward: q1/q2 = predictor(base_encoder(x1/x2)) (two passes: BatchNorm statistics
                                  per view), momentum update, k1/k2 = momentum_encoder(x1/x2) under no_grad,
                                  loss = ctr(q1, k2) + ctr(q2, k1);  ctr(q, k) = CE(normalize(q) . normalize(gather(k))^T / T,
                                  arange(N) + N*rank) * 2T
* passl/models/utils/averaged_model.py:69-188   update_parameters: first call copies the parameters, later calls
                                  p_avg = p_avg*(1-m_t) + p*m_t with m_t = end - (end - momentum)*(cos(pi*steps/max_steps)+1)/2
                                  (momentum = base_momentum = 0.99, end = 0: as written in the reference the momentum
                                  encoder FOLLOWS the base encoder closely at the start and freezes towards the end).
                                  Paddle's BatchNorm keeps `_mean` / `_variance` as non-trainable PARAMETERS (they are in
                                  named_parameters(), there are no buffers), so the running statistics are averaged like
                                  every weight - and then moved again by the momentum encoder's own train-mode forward
* passl/optimizer/adamw.py:50-140 paddle `adamw` op over every trainable parameter (weight_decay 0.1, no exclusion list in
                                  tasks/ssl/mocov3/configs/mocov3_vit_base_patch16_224_pt_in1k_4n32c_dp_fp16o1.yaml)

State = flat ``dict[str, Tensor]`` with the reference's state_dict names (``base_encoder.*``, ``predictor.*``); the momentum
encoder's copy uses the same names (reference: ``momentum_encoder.model.0.* / model.1.*``).  Linear weights are [in, out].

[Paddle-semantics] assumptions: nn.LayerNorm(epsilon=1e-6) biased variance; nn.GELU exact erf; BatchNorm1D momentum 0.9,
eps 1e-5, biased running variance, train-mode batch statistics also under no_grad; F.normalize eps 1e-12; CrossEntropyLoss
mean; paddle.meshgrid is 'ij'; default Linear init of the projector / predictor (not stated in the tree) taken as Xavier
uniform — initial values only, every parity test loads explicit state.
"""
import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

from . import mae as M

BN_MOMENTUM, BN_EPS = 0.9, 1e-5

VIT_B = dict(img_size=224, patch_size=16, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4.0, dim=256, mlp_dim=4096)
SMALL = dict(img_size=64, patch_size=16, embed_dim=128, depth=2, num_heads=4, mlp_ratio=4.0, dim=64, mlp_dim=256)


def sincos_position_embedding(embed_dim, h, w, temperature=10000.):
    """mocov3.py:69-91 (grid_w, grid_h = meshgrid(arange(w), arange(h)) in 'ij' order, flattened)."""
    grid_w, grid_h = torch.meshgrid(torch.arange(w, dtype=torch.float32), torch.arange(h, dtype=torch.float32),
                                    indexing='ij')
    assert embed_dim % 4 == 0
    pos_dim = embed_dim // 4
    omega = torch.arange(pos_dim, dtype=torch.float32) / pos_dim
    omega = 1. / (temperature ** omega)
    out_w = grid_w.flatten()[..., None] @ omega[None]
    out_h = grid_h.flatten()[..., None] @ omega[None]
    pos = torch.cat([torch.sin(out_w), torch.cos(out_w), torch.sin(out_h), torch.cos(out_h)], dim=1)[None]
    return torch.cat([torch.zeros(1, 1, embed_dim), pos], dim=1)


def mlp_spec(prefix, num_layers, input_dim, mlp_dim, output_dim):
    """_build_mlp: [(linear key, bn key, affine?, relu?)] with the reference's Sequential indices."""
    out, idx = [], 0
    for l in range(num_layers):
        d1 = input_dim if l == 0 else mlp_dim
        d2 = output_dim if l == num_layers - 1 else mlp_dim
        last = l == num_layers - 1
        out.append(dict(lin='%s.%d' % (prefix, idx), bn='%s.%d' % (prefix, idx + 1), d1=d1, d2=d2,
                        affine=not last, relu=not last))
        idx += 2 if last else 3
    return out


def init_state(gen, img_size=224, patch_size=16, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4.0, dim=256,
               mlp_dim=4096):
    st = OrderedDict()
    g = img_size // patch_size
    E = 'base_encoder.'
    st[E + 'pos_embed'] = sincos_position_embedding(embed_dim, g, g)
    st[E + 'cls_token'] = torch.randn(1, 1, embed_dim, generator=gen) * 1e-6
    val = math.sqrt(6. / float(3 * patch_size * patch_size + embed_dim))
    st[E + 'patch_embed.proj.weight'] = (torch.rand(embed_dim, 3, patch_size, patch_size, generator=gen) * 2 - 1) * val
    st[E + 'patch_embed.proj.bias'] = torch.zeros(embed_dim)

    def uni(shape, a):
        return (torch.rand(shape, generator=gen) * 2 - 1) * a

    def lin(name, cin, cout, a=None, bias=True):
        st[name + '.weight'] = uni((cin, cout), math.sqrt(6.0 / (cin + cout)) if a is None else a)
        if bias:
            st[name + '.bias'] = torch.zeros(cout)

    hid = int(embed_dim * mlp_ratio)
    for i in range(depth):
        p = E + 'blocks.%d' % i
        st[p + '.norm1.weight'], st[p + '.norm1.bias'] = torch.ones(embed_dim), torch.zeros(embed_dim)
        lin(p + '.attn.qkv', embed_dim, 3 * embed_dim, a=math.sqrt(6. / float(embed_dim + embed_dim)))
        lin(p + '.attn.proj', embed_dim, embed_dim)
        st[p + '.norm2.weight'], st[p + '.norm2.bias'] = torch.ones(embed_dim), torch.zeros(embed_dim)
        lin(p + '.mlp.fc1', embed_dim, hid)
        lin(p + '.mlp.fc2', hid, embed_dim)
    st[E + 'norm.weight'], st[E + 'norm.bias'] = torch.ones(embed_dim), torch.zeros(embed_dim)
    for spec in mlp_spec(E + 'head', 3, embed_dim, mlp_dim, dim) + mlp_spec('predictor', 2, dim, mlp_dim, dim):
        lin(spec['lin'], spec['d1'], spec['d2'], bias=False)
        if spec['affine']:
            st[spec['bn'] + '.weight'], st[spec['bn'] + '.bias'] = torch.ones(spec['d2']), torch.zeros(spec['d2'])
        st[spec['bn'] + '._mean'], st[spec['bn'] + '._variance'] = torch.zeros(spec['d2']), torch.ones(spec['d2'])
    return st


def is_buffer(k):
    return k.endswith('._mean') or k.endswith('._variance')


FROZEN = ('base_encoder.pos_embed', 'base_encoder.patch_embed.proj.weight', 'base_encoder.patch_embed.proj.bias')


def trainable_keys(st):
    """stop_grad_conv1=True (mocov3_vit_base_pretrain): the patch embedding is frozen; pos_embed is fixed."""
    return [k for k in st if not is_buffer(k) and k not in FROZEN]


def param_keys(st):
    """named_parameters() of Sequential(base_encoder, predictor): what the momentum average covers — every entry,
    the BatchNorm running statistics included (Paddle parameters with stop_gradient=True)."""
    return list(st)


# ------------------------------------------------------------------ forward
def vit_features(st, x, cfg):
    E, p = 'base_encoder.', cfg['patch_size']
    N = x.shape[0]
    t = F.conv2d(x, st[E + 'patch_embed.proj.weight'], st[E + 'patch_embed.proj.bias'], stride=p)
    t = t.flatten(2).transpose(1, 2)
    t = torch.cat([st[E + 'cls_token'].expand(N, -1, -1), t], dim=1) + st[E + 'pos_embed']
    for i in range(cfg['depth']):
        t = M.block_forward(st, E + 'blocks.%d' % i, t, cfg['num_heads'])
    t = M.layer_norm(t, st[E + 'norm.weight'], st[E + 'norm.bias'])
    return t[:, 0]


def batch_norm1d(st, key, x):
    """Train-mode BatchNorm1D over [N, C]; the running statistics in ``st`` are updated in place (no autograd)."""
    mean = x.mean(dim=0)
    var = x.var(dim=0, unbiased=False)
    with torch.no_grad():
        st[key + '._mean'] = BN_MOMENTUM * st[key + '._mean'] + (1 - BN_MOMENTUM) * mean.detach()
        st[key + '._variance'] = BN_MOMENTUM * st[key + '._variance'] + (1 - BN_MOMENTUM) * var.detach()
    inv = torch.rsqrt(var + BN_EPS)
    if key + '.weight' in st:
        return (x - mean[None]) * (inv * st[key + '.weight'])[None] + st[key + '.bias'][None]
    return (x - mean[None]) * inv[None]


def mlp_forward(st, specs, x):
    for s in specs:
        x = batch_norm1d(st, s['bn'], x @ st[s['lin'] + '.weight'])
        if s['relu']:
            x = F.relu(x)
    return x


def encode(st, x, cfg):
    """predictor(base_encoder(x)) of one view: nn.Sequential(base_encoder, predictor)."""
    feat = vit_features(st, x, cfg)
    proj = mlp_forward(st, mlp_spec('base_encoder.head', 3, cfg['embed_dim'], cfg['mlp_dim'], cfg['dim']), feat)
    return mlp_forward(st, mlp_spec('predictor', 2, cfg['dim'], cfg['mlp_dim'], cfg['dim']), proj)


def l2n(x, eps=1e-12):
    return x / x.norm(dim=1, keepdim=True).clamp_min(eps)


def contrastive_loss(q, k_all, T, rank=0):
    q, k_all = l2n(q), l2n(k_all)
    logits = q @ k_all.t() / T
    N = q.shape[0]
    labels = torch.arange(N) + N * rank
    return F.cross_entropy(logits, labels) * (2 * T), logits


class MoCoV3Oracle:
    def __init__(self, cfg=None, seed=0, T=0.2, base_momentum=0.99, max_steps=1000, lr=1.5e-4, beta1=0.9,
                 beta2=0.999, eps=1e-8, weight_decay=0.1, dtype=torch.float32):
        self.cfg = dict(VIT_B if cfg is None else cfg)
        gen = torch.Generator().manual_seed(seed)
        self.st = OrderedDict((k, v.to(dtype)) for k, v in init_state(gen, **self.cfg).items())
        # deepcopy(model) at construction (BaseAveragedModel.__init__)
        self.mom = OrderedDict((k, v.clone()) for k, v in self.st.items())
        self.T, self.momentum, self.end_momentum, self.max_steps = T, base_momentum, 0.0, max_steps
        self.steps = 0                       # momentum_encoder.steps
        self.lr_value, self.b1, self.b2, self.eps, self.wd = lr, beta1, beta2, eps, weight_decay
        self.m, self.v = OrderedDict(), OrderedDict()
        self.step_count = 0

    def lr(self):
        return self.lr_value(self.step_count) if callable(self.lr_value) else self.lr_value

    def ema_momentum(self):
        c = (math.cos(math.pi * self.steps / float(self.max_steps)) + 1) / 2
        return self.end_momentum - (self.end_momentum - self.momentum) * c

    @torch.no_grad()
    def update_momentum_encoder(self):
        if self.steps == 0:
            for k in param_keys(self.st):
                self.mom[k] = self.st[k].detach().clone()
        else:
            m = self.ema_momentum()
            for k in param_keys(self.st):
                self.mom[k] = self.mom[k] * (1.0 - m) + self.st[k].detach() * m
        self.steps += 1

    def forward_backward(self, x1, x2, k_gather=None, rank=0):
        """k_gather(k) -> keys of every rank (identity for one rank)."""
        tk = trainable_keys(self.st)
        for n in tk:
            self.st[n] = self.st[n].detach().requires_grad_(True)
        q1 = encode(self.st, x1, self.cfg)
        q2 = encode(self.st, x2, self.cfg)
        with torch.no_grad():
            self.update_momentum_encoder()
            k1 = encode(self.mom, x1, self.cfg)
            k2 = encode(self.mom, x2, self.cfg)
        g = k_gather or (lambda t: t)
        l1, logits1 = contrastive_loss(q1, g(k2), self.T, rank)
        l2, _ = contrastive_loss(q2, g(k1), self.T, rank)
        loss = l1 + l2
        loss.backward()
        grads = OrderedDict((n, self.st[n].grad.detach().clone()) for n in tk)
        return dict(loss=loss.detach(), q1=q1.detach(), q2=q2.detach(), k1=k1, k2=k2, logits1=logits1.detach(),
                    grads=grads)

    def train_step(self, x1, x2):
        out = self.forward_backward(x1, x2)
        self.apply_adamw(out['grads'])
        return out

    apply_adamw = M.MAEOracle.apply_adamw
