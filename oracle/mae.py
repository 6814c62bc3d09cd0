"""Oracle: one MAE pre-training step (ViT encoder/decoder, random masking, masked-patch MSE,
AdamW), torch-CPU fp32/fp64.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Follows:

* passl_v110/modeling/backbones/mae.py:318-564  class MAE (= passl/models/mae.py:37-290, the v2 twin):
  patch-embed conv 16x16/s16 -> + pos_embed[1:] -> random_masking (argsort of uniform noise, keep
  the len_keep smallest) -> cls token (+ pos_embed[:1]) -> `depth` pre-norm blocks -> norm;
  decoder_embed -> append mask tokens, unshuffle by ids_restore -> + decoder_pos_embed ->
  `decoder_depth` blocks -> decoder_norm -> decoder_pred -> drop cls; loss = masked mean of the
  per-patch MSE against (optionally per-patch normalised) pixels
* mae.py:61-189  Mlp (fc1-GELU-fc2), Attention (qkv Linear, softmax(q k^T * d^-0.5) v, proj), Block
* passl_v110/modules/get_sincos_pe.py:18-75  fixed 2-D sin-cos position embeddings
* passl_v110/modeling/architectures/MAE.py:30-55  MAE_PRETRAIN wrapper (its train_iter passes the
  whole input tuple and returns a tuple — unusable with OptimizerHook; the oracle uses inputs[0]
  and returns {'loss'} as SURVEY §3.4 prescribes)
* configs/mae/mae_vit_b_pretrain.yaml: AdamW(beta1 .9, beta2 .95, wd .05) over model.parameters()
  (no exclusion list: biases and LayerNorm affine are decayed too), LinearWarmup o CosineAnnealingDecay

[Paddle-semantics] assumptions: nn.LayerNorm(epsilon=1e-6) biased variance; nn.GELU = exact erf
form; Linear weight [in,out]; Tensor.var default unbiased (norm_pix_loss); argsort ascending,
stable for ties is irrelevant (continuous noise); AdamW (adamw op): p *= 1 - lr*wd;
m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2; p -= lr*sqrt(1-b2^t)/(1-b1^t) * m / (sqrt(v) + eps*sqrt(1-b2^t)),
eps 1e-8; parameters with stop_gradient (pos embeddings) are not updated.
"""
import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F


# ------------------------------------------------------------------ position embedding
def sincos_1d(embed_dim, pos):
    omega = np.arange(embed_dim // 2, dtype=np.float64)
    omega /= embed_dim / 2.
    omega = 1. / 10000 ** omega
    out = np.einsum('m,d->md', pos.reshape(-1), omega)
    return np.concatenate([np.sin(out), np.cos(out)], axis=1)


def sincos_2d(embed_dim, grid_size, cls_token=True):
    """get_2d_sincos_pos_embed (w goes first in the meshgrid)."""
    gh = np.arange(grid_size, dtype=np.float32)
    gw = np.arange(grid_size, dtype=np.float32)
    grid = np.stack(np.meshgrid(gw, gh), axis=0).reshape([2, 1, grid_size, grid_size])
    emb = np.concatenate([sincos_1d(embed_dim // 2, grid[0]), sincos_1d(embed_dim // 2, grid[1])], axis=1)
    if cls_token:
        emb = np.concatenate([np.zeros([1, embed_dim]), emb], axis=0)
    return emb


# ------------------------------------------------------------------ state
def _xavier_uniform(gen, fan_in, fan_out, shape):
    a = math.sqrt(6.0 / (fan_in + fan_out))
    return (torch.rand(shape, generator=gen) * 2 - 1) * a


def block_keys(prefix):
    return [prefix + s for s in ('.norm1.weight', '.norm1.bias', '.attn.qkv.weight', '.attn.qkv.bias',
                                 '.attn.proj.weight', '.attn.proj.bias', '.norm2.weight', '.norm2.bias',
                                 '.mlp.fc1.weight', '.mlp.fc1.bias', '.mlp.fc2.weight', '.mlp.fc2.bias')]


def init_state(gen, img_size=224, patch_size=16, in_chans=3, embed_dim=768, depth=12,
               decoder_embed_dim=512, decoder_depth=8, mlp_ratio=4.0):
    """Keys = the reference MAE backbone's state_dict names; Linear weights [in, out]."""
    st = OrderedDict()
    g = img_size // patch_size
    L = g * g
    pdim = in_chans * patch_size * patch_size
    st['cls_token'] = torch.fmod(torch.randn(1, 1, embed_dim, generator=gen), 2.0) * 0.02
    st['pos_embed'] = torch.tensor(sincos_2d(embed_dim, g), dtype=torch.float32).unsqueeze(0)
    st['mask_token'] = torch.fmod(torch.randn(1, 1, decoder_embed_dim, generator=gen), 2.0) * 0.02
    st['decoder_pos_embed'] = torch.tensor(sincos_2d(decoder_embed_dim, g), dtype=torch.float32).unsqueeze(0)
    st['patch_embed.proj.weight'] = _xavier_uniform(gen, pdim, embed_dim, (embed_dim, pdim)).reshape(
        embed_dim, in_chans, patch_size, patch_size)
    st['patch_embed.proj.bias'] = torch.zeros(embed_dim)

    def lin(name, cin, cout):
        st[name + '.weight'] = _xavier_uniform(gen, cin, cout, (cin, cout))
        st[name + '.bias'] = torch.zeros(cout)

    def ln(name, dim):
        st[name + '.weight'] = torch.ones(dim)
        st[name + '.bias'] = torch.zeros(dim)

    def blocks(prefix, n, dim):
        hid = int(dim * mlp_ratio)
        for i in range(n):
            p = '%s.%d' % (prefix, i)
            ln(p + '.norm1', dim)
            lin(p + '.attn.qkv', dim, 3 * dim)
            lin(p + '.attn.proj', dim, dim)
            ln(p + '.norm2', dim)
            lin(p + '.mlp.fc1', dim, hid)
            lin(p + '.mlp.fc2', hid, dim)
    blocks('blocks', depth, embed_dim)
    ln('norm', embed_dim)
    lin('decoder_embed', embed_dim, decoder_embed_dim)
    blocks('decoder_blocks', decoder_depth, decoder_embed_dim)
    ln('decoder_norm', decoder_embed_dim)
    lin('decoder_pred', decoder_embed_dim, pdim)
    return st


FROZEN = ('pos_embed', 'decoder_pos_embed')


def trainable_keys(st):
    return [k for k in st if k not in FROZEN]


# ------------------------------------------------------------------ forward
def layer_norm(x, w, b, eps=1e-6):
    mu = x.mean(dim=-1, keepdim=True)
    var = x.var(dim=-1, unbiased=False, keepdim=True)
    return (x - mu) * torch.rsqrt(var + eps) * w + b


def gelu(x):
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))


def block_forward(st, p, x, num_heads):
    B, T, C = x.shape
    d = C // num_heads
    h = layer_norm(x, st[p + '.norm1.weight'], st[p + '.norm1.bias'])
    qkv = (h @ st[p + '.attn.qkv.weight'] + st[p + '.attn.qkv.bias']).reshape(B, T, 3, num_heads, d)
    qkv = qkv.permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    attn = torch.softmax((q @ k.transpose(-1, -2)) * d ** -0.5, dim=-1)
    a = (attn @ v).permute(0, 2, 1, 3).reshape(B, T, C)
    x = x + (a @ st[p + '.attn.proj.weight'] + st[p + '.attn.proj.bias'])
    h = layer_norm(x, st[p + '.norm2.weight'], st[p + '.norm2.bias'])
    h = gelu(h @ st[p + '.mlp.fc1.weight'] + st[p + '.mlp.fc1.bias'])
    return x + (h @ st[p + '.mlp.fc2.weight'] + st[p + '.mlp.fc2.bias'])


def patchify(imgs, p):
    N, C, H, W = imgs.shape
    h = w = H // p
    x = imgs.reshape(N, C, h, p, w, p)
    x = torch.einsum('nchpwq->nhwpqc', x)
    return x.reshape(N, h * w, p * p * C)


def random_masking_ids(noise, mask_ratio):
    N, L = noise.shape
    len_keep = int(L * (1 - mask_ratio))
    ids_shuffle = torch.argsort(noise, dim=1)
    ids_restore = torch.argsort(ids_shuffle, dim=1)
    ids_keep = ids_shuffle[:, :len_keep]
    mask = torch.ones(N, L, dtype=noise.dtype)
    mask[:, :len_keep] = 0
    mask = torch.gather(mask, 1, ids_restore)
    return ids_keep, mask, ids_restore


def mae_forward(st, imgs, noise, cfg, mask_ratio=0.75):
    """Returns loss, pred [N,L,p*p*3], mask [N,L].  `noise` [N,L] replaces paddle.rand."""
    p, nh, dnh = cfg['patch_size'], cfg['num_heads'], cfg['decoder_num_heads']
    N = imgs.shape[0]
    x = F.conv2d(imgs, st['patch_embed.proj.weight'], st['patch_embed.proj.bias'], stride=p)
    x = x.flatten(2).transpose(1, 2)
    x = x + st['pos_embed'][:, 1:, :]
    ids_keep, mask, ids_restore = random_masking_ids(noise, mask_ratio)
    D = x.shape[-1]
    x = torch.gather(x, 1, ids_keep.unsqueeze(-1).expand(-1, -1, D))
    cls = (st['cls_token'] + st['pos_embed'][:, :1, :]).expand(N, -1, -1)
    x = torch.cat([cls, x], dim=1)
    for i in range(cfg['depth']):
        x = block_forward(st, 'blocks.%d' % i, x, nh)
    x = layer_norm(x, st['norm.weight'], st['norm.bias'])
    latent = x
    # decoder
    x = x @ st['decoder_embed.weight'] + st['decoder_embed.bias']
    L = ids_restore.shape[1]
    Dd = x.shape[-1]
    mask_tokens = st['mask_token'].expand(N, L + 1 - x.shape[1], -1)
    x_ = torch.cat([x[:, 1:, :], mask_tokens], dim=1)
    x_ = torch.gather(x_, 1, ids_restore.unsqueeze(-1).expand(-1, -1, Dd))
    x = torch.cat([x[:, :1, :], x_], dim=1)
    x = x + st['decoder_pos_embed']
    for i in range(cfg['decoder_depth']):
        x = block_forward(st, 'decoder_blocks.%d' % i, x, dnh)
    x = layer_norm(x, st['decoder_norm.weight'], st['decoder_norm.bias'])
    x = x @ st['decoder_pred.weight'] + st['decoder_pred.bias']
    pred = x[:, 1:, :]
    target = patchify(imgs, p)
    if cfg.get('norm_pix_loss', False):
        mean = target.mean(dim=-1, keepdim=True)
        var = target.var(dim=-1, keepdim=True)                # unbiased, like paddle's default
        target = (target - mean) / (var + 1.e-6) ** .5
    loss = ((pred - target) ** 2).mean(dim=-1)
    loss = (loss * mask).sum() / mask.sum()
    return loss, pred, mask, latent


# ------------------------------------------------------------------ solver
def finetune_state(keys_shapes, seed=0):
    """Seed-defined state for the fine-tuning model (MAE_FINETUNE: reference architectures/MAE.py:58-94): deterministic
    non-trivial values for every (key, shape) of the model's own state_dict, in key order — LayerNorm weights around 1,
    small non-zero biases, N(0, 0.02) elsewhere.  tests/golden/make_golden_mae_finetune.py loads it into the reference's
    model, tests/test_mae_gpu.py into the product's (the fixture records the key list)."""
    gen = torch.Generator().manual_seed(seed)
    st = {}
    for name, shape in keys_shapes:
        t = torch.randn(*shape, generator=gen)
        if name.endswith('.bias'):
            st[name] = 0.02 * t
        elif name.endswith('.weight') and 'norm' in (name.split('.')[-2] if name.count('.') else ''):
            st[name] = 1.0 + 0.1 * t
        elif name.endswith('cls_token') or name.endswith('pos_embed'):
            st[name] = 0.02 * t
        else:
            st[name] = 0.02 * t
    return st


def warmup_cosine_lr(t, base_lr, t_max, eta_min, warmup_steps, start_lr, end_lr):
    """LinearWarmup(learning_rate=CosineAnnealingDecay(base_lr, T_max, eta_min), warmup_steps,
    start_lr, end_lr) at scheduler epoch t (units already converted to iterations)."""
    if t < warmup_steps:
        return (end_lr - start_lr) * float(t) / float(warmup_steps) + start_lr
    tt = t - warmup_steps
    return eta_min + (base_lr - eta_min) * (1 + math.cos(math.pi * tt / t_max)) / 2


VIT_B = dict(img_size=224, patch_size=16, embed_dim=768, depth=12, num_heads=12,
             decoder_embed_dim=512, decoder_depth=8, decoder_num_heads=16, mlp_ratio=4.0)


class MAEOracle:
    def __init__(self, cfg=None, seed=0, lr=1e-3, beta1=0.9, beta2=0.95, eps=1e-8, weight_decay=0.05,
                 mask_ratio=0.75, dtype=torch.float32):
        self.cfg = dict(VIT_B if cfg is None else cfg)
        gen = torch.Generator().manual_seed(seed)
        arch = {k: self.cfg[k] for k in ('img_size', 'patch_size', 'embed_dim', 'depth',
                                         'decoder_embed_dim', 'decoder_depth', 'mlp_ratio')}
        self.st = OrderedDict((k, v.to(dtype)) for k, v in init_state(gen, **arch).items())
        self.lr_value, self.b1, self.b2, self.eps, self.wd = lr, beta1, beta2, eps, weight_decay
        self.mask_ratio = mask_ratio
        self.m, self.v = OrderedDict(), OrderedDict()
        self.step_count = 0

    def lr(self):
        return self.lr_value(self.step_count) if callable(self.lr_value) else self.lr_value

    def train_step(self, imgs, noise):
        tk = trainable_keys(self.st)
        for n in tk:
            self.st[n] = self.st[n].detach().requires_grad_(True)
        loss, pred, mask, latent = mae_forward(self.st, imgs, noise, self.cfg, self.mask_ratio)
        loss.backward()
        grads = OrderedDict((n, self.st[n].grad.detach().clone()) for n in tk)
        self.apply_adamw(grads)
        return dict(loss=loss.detach(), pred=pred.detach(), mask=mask, latent=latent.detach(), grads=grads)

    @torch.no_grad()
    def apply_adamw(self, grads):
        lr = self.lr()
        self.step_count += 1
        t = self.step_count
        b1p, b2p = self.b1 ** t, self.b2 ** t
        for n, g in grads.items():
            p = self.st[n].detach()
            m = self.m.get(n, torch.zeros_like(p))
            v = self.v.get(n, torch.zeros_like(p))
            p = p * (1.0 - lr * self.wd)
            m = self.b1 * m + (1 - self.b1) * g
            v = self.b2 * v + (1 - self.b2) * g * g
            lr_t = lr * math.sqrt(1 - b2p) / (1 - b1p)
            p = p - lr_t * (m / (v.sqrt() + self.eps * math.sqrt(1 - b2p)))
            self.m[n], self.v[n] = m, v
            self.st[n] = p.detach()
