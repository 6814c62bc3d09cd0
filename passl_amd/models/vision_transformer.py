"""``passl.models.vision_transformer`` — the v2 Vision Transformer on the MI355X HIP path.

Names, constructor arguments, sub-layer / state_dict keys and the factories are the reference's
(passl/models/vision_transformer.py: Mlp :84-113, Attention :116-156, Block :159-206, PatchEmbed :209-249,
VisionTransformer :252-430, factories :432-616).  No kernel of its own: the blocks are the MAE path's
(passl_amd/modeling/backbones/mae.py — tokens as 2-D rows [B*T, C] in the compute dtype, every Linear the
implicit-GEMM kernel with bias / residual epilogues, fused attention, LayerNorm / GELU kernels of csrc/vit.hip);
the learnable ``cls_token`` / ``pos_embed`` use the CLIP tower's class + position kernel pair
(passl_amd/modeling/backbones/vision_transformer.py:_ClsPosFn: the gradient of the position table is one column
sum); the classifier head works on fp32 rows.

Envelope: what the fused attention kernel covers (sequence <= 208 tokens, head dim 32 or 64) — i.e. the 224-pixel
base / large variants; the 384-pixel and 14-pixel-patch factories construct (names, shapes and state_dict are the
reference's) and raise PASSL_EUNSUPPORTED when run.  Dropout / stochastic depth / ``qk_scale`` are not built (the
pre-training recipes of tasks/ssl keep them at zero).

Initialisation [Paddle-semantics]: the reference relies on Paddle's defaults for the blocks' Linears (Xavier-uniform
weights, zero bias) and the patch convolution (Normal(0, sqrt(2 / fan_in_of_the_filter))) and sets explicitly:
``pos_embed`` ~ N(0, 0.02), ``cls_token`` = 0, LayerNorm (1, 0), ``head`` zeros (or Xavier / -10 bias with a
representation layer) — vision_transformer.py:318-337."""
import math
import os
import pickle
from functools import partial

import torch
import torch.nn as tnn

from ..hip import config, ops
from ..hip import nn as hnn
from ..hip.nn import EncoderArena
from ..modeling.backbones.mae import Attention, Block, Mlp, PatchEmbed          # noqa: F401  (reference names)
from ..modeling.backbones.vision_transformer import _ClsPosFn
from ..utils.checkpoint import load_lenient, load_pickle, to_numpy
from .base_model import Model

__all__ = [
    'ViT_base_patch16_224', 'ViT_base_patch16_384', 'ViT_base_patch32_224', 'ViT_base_patch32_384',
    'ViT_large_patch16_224', 'ViT_large_patch16_384', 'ViT_large_patch32_224', 'ViT_large_patch32_384',
    'ViT_huge_patch14_224', 'ViT_huge_patch14_384', 'ViT_g_patch14_224', 'ViT_G_patch14_224', 'ViT_6B_patch14_224',
    'VisionTransformer',
]


def to_2tuple(x):
    return tuple([x] * 2)


@torch.no_grad()
def _xavier_uniform(w, fan_in, fan_out):
    w.copy_((torch.rand(w.shape) * 2 - 1) * math.sqrt(6.0 / (fan_in + fan_out)))


class VisionTransformer(Model):
    """Vision Transformer with support for patch input."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, class_num=1000, embed_dim=768, depth=12,
                 num_heads=12, mlp_ratio=4, qkv_bias=False, qk_scale=None, drop_rate=0., attn_drop_rate=0.,
                 drop_path_rate=0., norm_layer='nn.LayerNorm', epsilon=1e-5, representation_size=None, **kwargs):
        super().__init__()
        if drop_rate or attn_drop_rate or drop_path_rate or qk_scale is not None:
            raise NotImplementedError('dropout / stochastic depth / qk_scale are not built on the HIP path (zero in '
                                      'the pre-training recipes)')
        dev = config.get_device()
        self.class_num = class_num
        self.representation_size = representation_size
        self.num_features = self.embed_dim = embed_dim
        if isinstance(norm_layer, str):
            if norm_layer != 'nn.LayerNorm':
                raise TypeError('The norm_layer must be str or paddle.nn.layer.Layer class')
            norm_layer = partial(hnn.LayerNorm, epsilon=epsilon)
        elif not callable(norm_layer):
            raise TypeError('The norm_layer must be str or paddle.nn.layer.Layer class')
        self.patch_embed = PatchEmbed(img_size=img_size, patch_size=patch_size, in_chans=in_chans,
                                      embed_dim=embed_dim)
        num_patches = self.patch_embed.num_patches
        # the attention kernels' envelope (csrc/attention.hip: head dimension 32 or 64, at most 208 tokens): a model
        # outside it would build and then fail at its first forward — say so at construction (the reference is
        # shape-generic: passl/models/vision_transformer.py:142-156)
        if embed_dim % num_heads or embed_dim // num_heads not in ops.ATTENTION_HEAD_DIMS or \
                num_patches + 1 > ops.ATTENTION_MAX_TOKENS:
            raise NotImplementedError(
                'VisionTransformer(img_size=%s, patch_size=%s, embed_dim=%d, num_heads=%d): %d tokens x head dimension '
                '%s is outside the HIP attention kernels (head dimension in %s, at most %d tokens; csrc/attention.hip) '
                '— 384^2 inputs and the huge / g / G / 6B widths need the key-tiled kernel that is not built'
                % (img_size, patch_size, embed_dim, num_heads, num_patches + 1,
                   embed_dim / float(num_heads), sorted(ops.ATTENTION_HEAD_DIMS), ops.ATTENTION_MAX_TOKENS))
        self.pos_embed = tnn.Parameter(torch.zeros(1, num_patches + 1, embed_dim, device=dev))
        self.cls_token = tnn.Parameter(torch.zeros(1, 1, embed_dim, device=dev))
        self.blocks = tnn.ModuleList([Block(embed_dim, num_heads, mlp_ratio, qkv_bias=qkv_bias, norm_layer=norm_layer)
                                      for _ in range(depth)])
        self.norm = norm_layer(embed_dim)
        # classifier head
        if representation_size is not None:
            self.head0 = hnn.Linear(embed_dim, representation_size)
            self.tanh = hnn.Tanh()
            self.head = hnn.Linear(representation_size, class_num) if class_num > 0 else None
        else:
            self.head = hnn.Linear(embed_dim, class_num) if class_num > 0 else None
        self._ids = {}
        with torch.no_grad():
            for name, m in self.named_modules():
                if isinstance(m, hnn.Linear):                       # Paddle's default for nn.Linear
                    _xavier_uniform(m.weight, m.weight.shape[0], m.weight.shape[1])
                    if m.bias is not None:
                        m.bias.zero_()
            w = self.patch_embed.proj.weight                        # Paddle's default for nn.Conv2D
            w.copy_(torch.randn(w.shape) * math.sqrt(2.0 / (w.shape[1] * w.shape[2] * w.shape[3])))
            self.patch_embed.proj.bias.zero_()
            if representation_size is not None:
                if self.head is not None:
                    self.head.bias.fill_(-10.0)
            elif self.head is not None:
                self.head.weight.zero_()
                self.head.bias.zero_()
            self.pos_embed.copy_(torch.randn(self.pos_embed.shape) * 0.02)
            self.cls_token.zero_()
        self.arena_q = EncoderArena(self, trainable=True)

    def sync_runtime_state(self):
        self.arena_q.refresh()

    def load_state_dict(self, state_dict, strict=True):
        r = super().load_state_dict(state_dict, strict=strict)
        self.sync_runtime_state()
        return r

    def _identity_ids(self, B, L, device):
        key = (B, L)
        if key not in self._ids:
            self._ids[key] = (torch.arange(L, dtype=torch.int32, device=device).repeat(B, 1).contiguous(),
                              (torch.arange(B, dtype=torch.int32, device=device) * (L + 1)).contiguous())
        return self._ids[key]

    def forward_features(self, x):
        B = x.shape[0]
        L = self.patch_embed.num_patches
        x = self.patch_embed(x)                                           # [B*L, D]
        ids, cls_rows = self._identity_ids(B, L, x.device)
        x = _ClsPosFn.apply(x, self.cls_token, self.pos_embed, ids, B, L)     # concat(cls, x) + pos_embed
        for blk in self.blocks:
            x = blk(x, B, L + 1)
        return self.norm(hnn.gather_rows(x, cls_rows))                    # norm(x)[:, 0]  (LayerNorm is per token)

    def forward(self, x):
        x = self.forward_features(x)
        if self.representation_size is not None:
            x = self.tanh(self.head0(x))
        return x if self.head is None else self.head(x, out_f32=True)

    # ---- vision_transformer.py:365-430
    def load_pretrained(self, path, rank=0, finetune=False):
        if not os.path.exists(path + '.pdparams'):
            raise ValueError('Model pretrain path {} does not exists.'.format(path))
        sd = load_pickle(path + '.pdparams')
        if finetune:
            for k in ['head0.weight', 'head0.bias', 'head.weight', 'head.bias']:
                sd.pop(k, None)
            pos = torch.as_tensor(sd['pos_embed']).float()
            n_new = self.patch_embed.num_patches
            extra = self.pos_embed.shape[-2] - n_new
            orig, new = int((pos.shape[-2] - extra) ** 0.5), int(n_new ** 0.5)
            if orig != new:                  # bicubic interpolation of the position tokens, class token unchanged
                tok = pos[0, extra:].reshape(1, orig, orig, -1).permute(0, 3, 1, 2)
                tok = torch.nn.functional.interpolate(tok, size=(new, new), mode='bicubic', align_corners=False)
                sd['pos_embed'] = torch.cat([pos[:, :extra], tok.permute(0, 2, 3, 1).flatten(1, 2)], dim=1).numpy()
        load_lenient(self, sd, what='pretrained ViT')
        self.sync_runtime_state()

    def save(self, path, local_rank=0, rank=0):
        if rank != 0:
            return
        os.makedirs(os.path.dirname(path) or '.', exist_ok=True)
        with open(path + '.pdparams', 'wb') as f:
            pickle.dump(to_numpy(dict(self.state_dict())), f, protocol=2)


def ViT_base_patch16_224(**kwargs):
    return VisionTransformer(patch_size=16, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4, qkv_bias=True,
                             epsilon=1e-6, representation_size=768, **kwargs)


def ViT_base_patch16_384(**kwargs):
    return VisionTransformer(img_size=384, patch_size=16, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4,
                             qkv_bias=True, epsilon=1e-6, representation_size=None, **kwargs)


def ViT_base_patch32_224(**kwargs):
    return VisionTransformer(patch_size=32, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4, qkv_bias=True,
                             epsilon=1e-6, representation_size=768, **kwargs)


def ViT_base_patch32_384(**kwargs):
    return VisionTransformer(img_size=384, patch_size=32, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4,
                             qkv_bias=True, epsilon=1e-6, representation_size=None, **kwargs)


def ViT_large_patch16_224(**kwargs):
    return VisionTransformer(patch_size=16, embed_dim=1024, depth=24, num_heads=16, mlp_ratio=4, qkv_bias=True,
                             epsilon=1e-6, representation_size=1024, **kwargs)


def ViT_large_patch16_384(**kwargs):
    return VisionTransformer(img_size=384, patch_size=16, embed_dim=1024, depth=24, num_heads=16, mlp_ratio=4,
                             qkv_bias=True, epsilon=1e-6, representation_size=None, **kwargs)


def ViT_large_patch32_224(**kwargs):
    return VisionTransformer(patch_size=32, embed_dim=1024, depth=24, num_heads=16, mlp_ratio=4, qkv_bias=True,
                             epsilon=1e-6, representation_size=1024, **kwargs)


def ViT_large_patch32_384(**kwargs):
    return VisionTransformer(img_size=384, patch_size=32, embed_dim=1024, depth=24, num_heads=16, mlp_ratio=4,
                             qkv_bias=True, epsilon=1e-6, representation_size=None, **kwargs)


def ViT_huge_patch14_224(**kwargs):
    return VisionTransformer(patch_size=14, embed_dim=1280, depth=32, num_heads=16, mlp_ratio=4,
                             representation_size=1280, **kwargs)


def ViT_huge_patch14_384(**kwargs):
    return VisionTransformer(img_size=384, patch_size=14, embed_dim=1280, depth=32, num_heads=16, mlp_ratio=4,
                             representation_size=None, **kwargs)


def ViT_g_patch14_224(**kwargs):
    return VisionTransformer(img_size=224, patch_size=14, embed_dim=1408, depth=40, num_heads=16, mlp_ratio=4.364,
                             qkv_bias=True, epsilon=1e-6, representation_size=1408, **kwargs)


def ViT_G_patch14_224(**kwargs):
    return VisionTransformer(img_size=224, patch_size=14, embed_dim=1664, depth=48, num_heads=16, mlp_ratio=4.9231,
                             qkv_bias=True, epsilon=1e-6, representation_size=1664, **kwargs)


def ViT_6B_patch14_224(**kwargs):
    return VisionTransformer(img_size=224, patch_size=14, embed_dim=2320, depth=80, num_heads=16, mlp_ratio=4.955,
                             qkv_bias=True, epsilon=1e-6, representation_size=2320, **kwargs)
