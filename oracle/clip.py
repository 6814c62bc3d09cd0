"""Oracle: one CLIP pre-training step (ViT image tower, causal text transformer, symmetric
cross-entropy over the scaled cosine similarities, AdamW), torch-CPU fp32/fp64.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Follows:

* passl_v110/modeling/backbones/clip.py:183-332  class CLIP: visual = VisionTransformer(pre_norm, proj),
  transformer = Transformer(attn_mask = triu(-inf, 1)), token_embedding, positional_embedding,
  ln_final, text_projection, logit_scale = ln(1/0.07); encode_text takes the row of the highest
  token id (EOT); features are divided by their L2 norm (no epsilon); image_logits =
  (exp(s) I) T^T, text_logits = (exp(s) T) I^T; logit_scale clipped in place to [-4.6, 4.6]
  after the logits are formed
* passl_v110/modeling/backbones/vision_transformer.py:69-225 Mlp / Attention (additive mask before
  the softmax) / Block (pre-norm, QuickGELU = x sigmoid(1.702 x), LayerNorm eps 1e-5) / Transformer;
  :267-366 VisionTransformer (patch conv without bias, class_embedding concat, + positional_embedding,
  norm_pre, blocks, norm_post(x[:, 0]) @ proj)
* passl_v110/modeling/architectures/CLIPWrapper.py:44-50 labels = arange(B); heads/clip_head.py:24-36
  loss = CE(image_logits) + CE(text_logits) (a sum, not a mean of the two)
* configs/clip/vit-b-32.yaml: AdamW(beta1 .9, beta2 .98, eps 1e-8, wd 5e-4) over every parameter,
  LinearWarmup o CosineAnnealingDecay.  The CLIP ctor's `qkv_bias` argument is not forwarded: both
  towers use the blocks' default (True).

[Paddle-semantics] assumptions: as oracle/mae.py (LayerNorm biased variance, Linear [in,out], AdamW)
plus: nn.Embedding = row gather, gradient = scatter-add; Tensor.argmax returns the first maximum;
Tensor.norm(axis=-1) = sqrt(sum x^2); softmax of a row with -inf entries gives exact zeros there;
CrossEntropyLoss = mean over the batch of logsumexp - logit[label].
"""
import math
from collections import OrderedDict

import torch
import torch.nn.functional as F

from .mae import layer_norm as _ln, warmup_cosine_lr  # noqa: F401  (same LayerNorm / schedule restatement)

VIT_B_32 = dict(embed_dim=512, image_resolution=224, vision_layers=12, vision_width=768,
                vision_patch_size=32, context_length=77, vocab_size=49408, transformer_width=512,
                transformer_heads=8, transformer_layers=12)
# BASELINE.json configs[4]: "CLIP ViT-B/16" = configs/clip/vit-b-32.yaml with vision_patch_size 16
# (197 image tokens; SURVEY appendix C: the reference ships only the B/32 yaml)
VIT_B_16 = dict(VIT_B_32, vision_patch_size=16)
SMALL = dict(embed_dim=64, image_resolution=64, vision_layers=2, vision_width=128,
             vision_patch_size=32, context_length=12, vocab_size=300, transformer_width=128,
             transformer_heads=2, transformer_layers=2)


def layer_norm(x, w, b):
    return _ln(x, w, b, eps=1e-5)


def quick_gelu(x):
    return x * torch.sigmoid(1.702 * x)


def _tn(gen, shape, std=0.02):
    return torch.fmod(torch.randn(shape, generator=gen), 2.0) * std


def _blocks(st, gen, prefix, n, dim, stds=None):
    for i in range(n):
        p = '%s.%d' % (prefix, i)
        for nm, cin, cout, key in (('.attn.qkv', dim, 3 * dim, 'attn'), ('.attn.proj', dim, dim, 'proj'),
                                   ('.mlp.fc1', dim, 4 * dim, 'fc'), ('.mlp.fc2', 4 * dim, dim, 'proj')):
            if stds is None:
                st[p + nm + '.weight'] = _tn(gen, (cin, cout))
            else:
                st[p + nm + '.weight'] = torch.randn(cin, cout, generator=gen) * stds[key]
            st[p + nm + '.bias'] = torch.zeros(cout)
        for nm in ('.norm1', '.norm2'):
            st[p + nm + '.weight'] = torch.ones(dim)
            st[p + nm + '.bias'] = torch.zeros(dim)


def init_state(gen, cfg, text_std_cap=None):
    """Keys = the reference CLIP backbone's state_dict names.  Initial distributions follow
    clip.py:251-282 / vision_transformer.py:331-342 (the text blocks' proj std is
    width^-0.5 * (2 depth), as written there); `text_std_cap` bounds it for the small goldens."""
    st = OrderedDict()
    w, p = cfg['vision_width'], cfg['vision_patch_size']
    L = (cfg['image_resolution'] // p) ** 2
    st['visual.class_embedding'] = _tn(gen, (1, 1, w))
    st['visual.positional_embedding'] = _tn(gen, (1, L + 1, w))
    st['visual.proj'] = torch.randn(w, cfg['embed_dim'], generator=gen) * w ** -0.5
    # nn.Conv2D default initializer: Normal(0, sqrt(2 / (k*k*in_channels)))
    st['visual.patch_embed.proj.weight'] = torch.randn(w, 3, p, p, generator=gen) * math.sqrt(2.0 / (3 * p * p))
    for nm in ('visual.norm_pre', 'visual.norm_post'):
        st[nm + '.weight'] = torch.ones(w)
        st[nm + '.bias'] = torch.zeros(w)
    _blocks(st, gen, 'visual.blocks', cfg['vision_layers'], w)
    tw, depth = cfg['transformer_width'], cfg['transformer_layers']
    proj_std = tw ** -0.5 * (2 * depth)
    if text_std_cap is not None:
        proj_std = min(proj_std, text_std_cap)
    _blocks(st, gen, 'transformer.blocks', depth, tw,
            stds=dict(proj=proj_std, attn=tw ** -0.5, fc=(2 * tw) ** -0.5))
    st['token_embedding.weight'] = torch.randn(cfg['vocab_size'], tw, generator=gen) * 0.02
    st['positional_embedding'] = torch.randn(cfg['context_length'], tw, generator=gen) * 0.01
    st['ln_final.weight'] = torch.ones(tw)
    st['ln_final.bias'] = torch.zeros(tw)
    st['text_projection'] = torch.randn(tw, cfg['embed_dim'], generator=gen) * tw ** -0.5
    st['logit_scale'] = torch.full((1,), math.log(1 / 0.07))
    return st


def make_text(gen, B, context_length, vocab_size):
    """Synthetic token rows: random ids, the EOT (= largest id, vocab_size-1) at a random position,
    zero padding after it (what the reference's tokenizer produces in shape and ordering)."""
    text = torch.zeros(B, context_length, dtype=torch.int64)
    for b in range(B):
        n = int(torch.randint(3, context_length + 1, (1,), generator=gen))
        text[b, :n - 1] = torch.randint(1, vocab_size - 2, (n - 1,), generator=gen)
        text[b, n - 1] = vocab_size - 1
    return text


def block_forward(st, p, x, num_heads, mask=None):
    B, T, C = x.shape
    d = C // num_heads
    h = layer_norm(x, st[p + '.norm1.weight'], st[p + '.norm1.bias'])
    qkv = (h @ st[p + '.attn.qkv.weight'] + st[p + '.attn.qkv.bias']).reshape(B, T, 3, num_heads, d)
    qkv = qkv.permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    attn = (q @ k.transpose(-1, -2)) * d ** -0.5
    if mask is not None:
        attn = attn + mask
    attn = torch.softmax(attn, dim=-1)
    a = (attn @ v).permute(0, 2, 1, 3).reshape(B, T, C)
    x = x + (a @ st[p + '.attn.proj.weight'] + st[p + '.attn.proj.bias'])
    h = layer_norm(x, st[p + '.norm2.weight'], st[p + '.norm2.bias'])
    h = quick_gelu(h @ st[p + '.mlp.fc1.weight'] + st[p + '.mlp.fc1.bias'])
    return x + (h @ st[p + '.mlp.fc2.weight'] + st[p + '.mlp.fc2.bias'])


def encode_image(st, image, cfg):
    w, p = cfg['vision_width'], cfg['vision_patch_size']
    B = image.shape[0]
    x = F.conv2d(image, st['visual.patch_embed.proj.weight'], None, stride=p).flatten(2).transpose(1, 2)
    x = torch.cat([st['visual.class_embedding'].expand(B, -1, -1), x], dim=1)
    x = x + st['visual.positional_embedding']
    x = layer_norm(x, st['visual.norm_pre.weight'], st['visual.norm_pre.bias'])
    for i in range(cfg['vision_layers']):
        x = block_forward(st, 'visual.blocks.%d' % i, x, w // 64)
    x = layer_norm(x[:, 0, :], st['visual.norm_post.weight'], st['visual.norm_post.bias'])
    return x @ st['visual.proj']


def encode_text(st, text, cfg):
    T = cfg['context_length']
    x = st['token_embedding.weight'][text] + st['positional_embedding']
    mask = torch.triu(torch.full((T, T), -math.inf, dtype=x.dtype), 1)
    for i in range(cfg['transformer_layers']):
        x = block_forward(st, 'transformer.blocks.%d' % i, x, cfg['transformer_heads'], mask)
    x = layer_norm(x, st['ln_final.weight'], st['ln_final.bias'])
    x = x[torch.arange(x.shape[0]), text.argmax(dim=-1)]
    return x @ st['text_projection']


def clip_forward(st, image, text, cfg):
    fi, ft = encode_image(st, image, cfg), encode_text(st, text, cfg)
    ni = fi / fi.norm(dim=-1, keepdim=True)
    nt = ft / ft.norm(dim=-1, keepdim=True)
    scale = st['logit_scale'].exp()
    image_logits = (scale * ni) @ nt.t()
    text_logits = (scale * nt) @ ni.t()
    labels = torch.arange(image.shape[0])
    img_loss = F.cross_entropy(image_logits, labels)
    text_loss = F.cross_entropy(text_logits, labels)
    return dict(loss=img_loss + text_loss, img_loss=img_loss, text_loss=text_loss,
                image_logits=image_logits, text_logits=text_logits, image_features=fi, text_features=ft)


class CLIPOracle:
    def __init__(self, cfg=None, seed=0, lr=1e-4, beta1=0.9, beta2=0.98, eps=1e-8, weight_decay=0.0005,
                 dtype=torch.float32, text_std_cap=None):
        self.cfg = dict(VIT_B_32 if cfg is None else cfg)
        gen = torch.Generator().manual_seed(seed)
        self.st = OrderedDict((k, v.to(dtype)) for k, v in init_state(gen, self.cfg, text_std_cap).items())
        self.lr_value, self.b1, self.b2, self.eps, self.wd = lr, beta1, beta2, eps, weight_decay
        self.m, self.v = OrderedDict(), OrderedDict()
        self.step_count = 0

    def lr(self):
        return self.lr_value(self.step_count) if callable(self.lr_value) else self.lr_value

    def train_step(self, image, text):
        for n in self.st:
            self.st[n] = self.st[n].detach().requires_grad_(True)
        out = clip_forward(self.st, image, text, self.cfg)
        out['loss'].backward()
        grads = OrderedDict((n, self.st[n].grad.detach().clone()) for n in self.st)
        with torch.no_grad():                                   # clip.py:309-311 (after the logits)
            self.st['logit_scale'] = self.st['logit_scale'].detach().clamp(-4.6, 4.6)
        self.apply_adamw(grads)
        res = {k: v.detach() for k, v in out.items()}
        res['grads'] = grads
        return res

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
