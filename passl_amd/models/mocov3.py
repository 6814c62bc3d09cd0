"""MoCo-v3 (ViT) pre-training on the MI355X HIP path — reference passl/models/mocov3.py.

Constructor arguments, factory names, sub-layer / state_dict names and the ``model([x1, x2]) -> loss`` contract are
the reference's: ``MoCoV3ViT`` :36-91 (VisionTransformer + FIXED 2-D sin-cos position embedding, frozen patch
embedding with ``stop_grad_conv1``), ``MoCoV3Pretrain`` :111-222 (``base_encoder.head`` replaced by the 3-layer
projector, 2-layer ``predictor``, ``momentum_encoder`` = CosineEMA over Sequential(base_encoder, predictor),
symmetric cross-rank InfoNCE), factories :260-296.

Execution (same building blocks as the MAE / CLIP paths, no new kernels):
  * tokens are 2-D rows [B*T, C] in the compute dtype; Linears are the implicit-GEMM kernels with bias / residual
    epilogues, LayerNorm / GELU / attention / token assembly are the kernels of csrc/vit.hip + csrc/attention.hip;
  * the trunk runs ONCE over both views (2N images: every op before the heads is per-sample, so batching the views
    changes nothing but the GEMM sizes), the BatchNorm MLPs run per view as in the reference (their statistics are
    per view);
  * projector / predictor: Linear accumulates and writes fp32, BatchNorm1D works on fp32 rows (SimCLR neck pattern);
  * all trainable state lives in ONE flat fp32 arena (``arena_q``: one AdamW launch, one bucketed all-reduce), the
    momentum encoder in a second arena of the same layout (``arena_k``): the cosine-momentum average is ONE
    launch over weights AND BatchNorm running statistics — Paddle keeps ``_mean`` / ``_variance`` as non-trainable
    parameters, so the reference's ``named_parameters()`` average covers them too (averaged_model.py:44-47, 72-80);
  * the frozen patch embedding has a non-trainable arena of its own and is SHARED by the momentum encoder (the
    reference's copy is `p*(1-m) + p*m` of an identical tensor: equal up to one rounding);
  * keys of every rank: all_gather of the normalised keys (no gradient), labels ``arange(N) + N*rank``.
"""
import math
import os
import pickle
from functools import partial

import torch
import torch.distributed as dist
import torch.nn as tnn
from torch.autograd import Function

from ..core.sync_utils import collectives_active
from ..hip import config, ops
from ..hip import nn as hnn
from ..hip.nn import EncoderArena
from ..modeling.backbones.mae import Block, PatchEmbed, _GatherFn
from ..modeling.heads.clip_head import _RowCEFn
from ..utils.checkpoint import load_lenient, load_pickle, to_numpy
from ..utils.infohub import runtime_info_hub
from .base_model import Model
from .utils.averaged_model import CosineEMA

__all__ = ['MoCoV3ViT', 'MoCoV3LinearProbe', 'MoCoV3Pretrain', 'mocov3_vit_base', 'mocov3_vit_base_linearprobe',
           'mocov3_vit_base_pretrain']


def build_2d_sincos_position_embedding(embed_dim, h, w, temperature=10000.):
    """mocov3.py:69-91: meshgrid(arange(w), arange(h)) in 'ij' order, [sin w, cos w, sin h, cos h], zero class row."""
    grid_w, grid_h = torch.meshgrid(torch.arange(w, dtype=torch.float32), torch.arange(h, dtype=torch.float32),
                                    indexing='ij')
    assert embed_dim % 4 == 0, 'Embed dimension must be divisible by 4 for 2D sin-cos position embedding'
    pos_dim = embed_dim // 4
    omega = 1. / (temperature ** (torch.arange(pos_dim, dtype=torch.float32) / pos_dim))
    out_w = grid_w.flatten()[:, None] * omega[None]
    out_h = grid_h.flatten()[:, None] * omega[None]
    pos = torch.cat([torch.sin(out_w), torch.cos(out_w), torch.sin(out_h), torch.cos(out_h)], dim=1)[None]
    return torch.cat([torch.zeros(1, 1, embed_dim), pos], dim=1)


@torch.no_grad()
def _uniform(w, a):
    w.copy_((torch.rand(w.shape) * 2 - 1) * a)


class _MLP(tnn.Sequential):
    """_build_mlp (mocov3.py:135-157): Linear(no bias) - BatchNorm1D - ReLU ... Linear(no bias) - BatchNorm1D(no
    gamma / beta), with the reference's Sequential indices as sub-layer names."""

    def forward(self, x):
        mods = list(self)
        dt = config.get_compute_dtype()
        i = 0
        while i < len(mods):
            relu = i + 2 < len(mods) and isinstance(mods[i + 2], hnn.ReLU)
            x = mods[i + 1](mods[i](hnn.to_compute(x, dt), out_f32=True), relu=relu)
            i += 3 if relu else 2
        return x                                                        # fp32 [N, dim]


def build_mlp(num_layers, input_dim, mlp_dim, output_dim, last_bn=True):
    mlp = []
    for l in range(num_layers):
        dim1 = input_dim if l == 0 else mlp_dim
        dim2 = output_dim if l == num_layers - 1 else mlp_dim
        lin = hnn.Linear(dim1, dim2, bias_attr=False)
        _uniform(lin.weight, math.sqrt(6.0 / (dim1 + dim2)))
        mlp.append(lin)
        if l < num_layers - 1:
            mlp.append(hnn.BatchNorm1D(dim2))
            mlp.append(hnn.ReLU())
        elif last_bn:
            mlp.append(hnn.BatchNorm1D(dim2, weight_attr=False, bias_attr=False))
        else:
            raise NotImplementedError('an MLP that ends in a Linear (last_bn=False) is not used by MoCo-v3')
    return _MLP(*mlp)


class MoCoV3ViT(hnn.Layer):
    def __init__(self, img_size=224, patch_size=16, in_chans=3, class_num=1000, embed_dim=768, depth=12,
                 num_heads=12, mlp_ratio=4, qkv_bias=False, qk_scale=None, drop_rate=0., attn_drop_rate=0.,
                 drop_path_rate=0., norm_layer='nn.LayerNorm', epsilon=1e-5, representation_size=None,
                 stop_grad_conv1=False, **kwargs):
        super().__init__()
        if drop_rate or attn_drop_rate or drop_path_rate or qk_scale is not None or representation_size is not None:
            raise NotImplementedError('dropout / stochastic depth / qk_scale / representation_size are not used by '
                                      'the MoCo-v3 pre-training recipe')
        dev = config.get_device()
        self.class_num = class_num
        self.num_features = self.embed_dim = embed_dim
        self.stop_grad_conv1 = bool(stop_grad_conv1)
        if isinstance(norm_layer, str):
            if norm_layer != 'nn.LayerNorm':
                raise NotImplementedError('norm_layer %r' % norm_layer)
            norm_layer = partial(hnn.LayerNorm, epsilon=epsilon)
        self.patch_embed = PatchEmbed(img_size=img_size, patch_size=patch_size, in_chans=in_chans,
                                      embed_dim=embed_dim)
        num_patches = self.patch_embed.num_patches
        g = self.patch_embed.grid_size
        # fixed 2-D sin-cos embedding (reference: a parameter with stop_gradient=True; same state_dict key)
        self.register_buffer('pos_embed', build_2d_sincos_position_embedding(embed_dim, g[0], g[1]).to(dev))
        self.cls_token = tnn.Parameter(torch.zeros(1, 1, embed_dim, device=dev))
        self.blocks = tnn.ModuleList([Block(embed_dim, num_heads, mlp_ratio, qkv_bias=qkv_bias, norm_layer=norm_layer)
                                      for _ in range(depth)])
        self.norm = norm_layer(embed_dim)
        self.head = hnn.Linear(embed_dim, class_num) if class_num > 0 else None
        self._ids = {}
        with torch.no_grad():
            for name, m in self.named_modules():
                if isinstance(m, hnn.Linear):
                    if 'qkv' in name:      # treat the weights of Q, K, V separately
                        _uniform(m.weight, math.sqrt(6. / float(m.weight.shape[1] // 3 + m.weight.shape[0])))
                    else:
                        _uniform(m.weight, math.sqrt(6. / float(m.weight.shape[0] + m.weight.shape[1])))
                    if m.bias is not None:
                        m.bias.zero_()
            self.cls_token.copy_(torch.randn(self.cls_token.shape) * 1e-6)
            w = self.patch_embed.proj.weight
            _uniform(w, math.sqrt(6. / float(3 * patch_size * patch_size + embed_dim)))
            self.patch_embed.proj.bias.zero_()
        if stop_grad_conv1:
            self.patch_embed.proj.weight.requires_grad_(False)
            self.patch_embed.proj.bias.requires_grad_(False)

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
        x = _GatherFn.apply(x, self.cls_token, self.pos_embed, ids, ids, B, L)     # [cls | patches] + pos_embed
        for blk in self.blocks:
            x = blk(x, B, L + 1)
        return self.norm(hnn.gather_rows(x, cls_rows))                    # norm(x)[:, 0]  (LayerNorm is per token)

    def forward(self, x):
        x = self.forward_features(x)
        return x if self.head is None else self.head(x)


class MoCoV3LinearProbe(MoCoV3ViT, Model):
    """mocov3.py:94-109: a MoCoV3ViT whose every parameter but ``head.weight`` / ``head.bias`` is frozen; the head
    starts at Normal(0, 0.01) / zero bias.  ``load_pretrained`` (VisionTransformer.load_pretrained,
    vision_transformer.py:365-381, finetune=False) takes the ``<prefix>_base_encoder.pdparams`` file
    ``MoCoV3Pretrain.save`` writes (backbone keys without prefix, no head).

    Execution: the frozen encoder lives in a NON-trainable arena and runs under no_grad (nothing is kept for
    backward); the head is the only trainable arena — one GEMM with fp32 scores."""

    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        # freeze all layers but the last fc
        for name, param in self.named_parameters():
            if name not in ['head.weight', 'head.bias']:
                param.requires_grad_(False)
        # optimize only the linear classifier
        parameters = [p for p in self.parameters() if p.requires_grad]
        assert len(parameters) == 2  # weight, bias
        with torch.no_grad():
            self.head.weight.copy_(torch.randn(self.head.weight.shape) * 0.01)
            self.head.bias.zero_()
        frozen = [self.patch_embed, self.blocks, self.norm]
        self.arena_k = EncoderArena(tnn.ModuleList(frozen), trainable=False)
        self.arena_q = EncoderArena(self.head, trainable=True)

    def sync_runtime_state(self):
        self.arena_k.refresh()
        self.arena_q.refresh()

    def load_state_dict(self, state_dict, strict=True):
        r = super().load_state_dict(state_dict, strict=strict)
        self.sync_runtime_state()
        return r

    def load_pretrained(self, path, rank=0, finetune=False):
        if not os.path.exists(path + '.pdparams'):
            raise ValueError('Model pretrain path {} does not exists.'.format(path))
        if finetune:
            raise NotImplementedError('finetune=True (head removal + position-embedding interpolation) belongs to the '
                                      'fine-tuning recipe')
        load_lenient(self, load_pickle(path + '.pdparams'), what='pretrained model')
        self.sync_runtime_state()

    def save(self, path, local_rank=0, rank=0):
        if rank != 0:
            return
        os.makedirs(os.path.dirname(path) or '.', exist_ok=True)
        with open(path + '.pdparams', 'wb') as f:
            pickle.dump(to_numpy(dict(self.state_dict())), f, protocol=2)

    def forward(self, x):
        self.arena_q.refresh()
        with torch.no_grad():
            feats = self.forward_features(x)
        return self.head(feats, out_f32=True)


class _KeyLogitsFn(Function):
    """logits = normalize(q) . k_all^T * (1/T) with k_all already normalised and constant (mocov3.py:170-176)."""

    @staticmethod
    def forward(ctx, q, k_all, alpha):
        q_n, norm = ops.l2norm_fwd(q.contiguous(), 1e-12)
        ctx.save_for_backward(q_n, norm, k_all, alpha)
        return ops.gemm_f32_nt(q_n, k_all, alpha)

    @staticmethod
    def backward(ctx, dlogits):
        q_n, norm, k_all, alpha = ctx.saved_tensors
        dq_n = ops.gemm_f32_gx(dlogits.contiguous(), k_all, alpha)
        return ops.l2norm_bwd(dq_n, q_n, norm, torch.float32), None, None


def concat_all_gather(t):
    """mocov3.py:161-168 (identity for one rank)."""
    if not collectives_active():
        return t
    out = torch.empty((dist.get_world_size() * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(out, t.contiguous())
    return out


class MoCoV3Pretrain(Model):
    """Build a MoCo model with a base encoder, a momentum encoder, and two MLPs."""

    def __init__(self, base_encoder, dim=256, mlp_dim=4096, T=1.0, base_momentum=0.01):
        super().__init__()
        self.T = T
        dev = config.get_device()
        self.base_encoder = base_encoder(class_num=mlp_dim)
        self.predictor = None
        self._build_projector_and_predictor_mlps(self, dim, mlp_dim)
        frozen = [self.base_encoder.patch_embed.proj] if self.base_encoder.stop_grad_conv1 else []
        # the momentum encoder: a structural twin (the reference deep-copies at construction)
        twin = hnn.Layer()
        twin.base_encoder = base_encoder(class_num=mlp_dim)
        self._build_projector_and_predictor_mlps(twin, dim, mlp_dim)
        if frozen:
            twin.base_encoder.patch_embed = self.base_encoder.patch_embed
        self.momentum_encoder = CosineEMA(tnn.Sequential(twin.base_encoder, twin.predictor), momentum=base_momentum)
        pair_q = tnn.ModuleList([self.base_encoder, self.predictor])      # the arena walk only: not a sub-layer
        self.arena_q = EncoderArena(pair_q, trainable=True, exclude=frozen)
        self.arena_k = EncoderArena(self.momentum_encoder.model, trainable=False, exclude=frozen)
        self.arena_pe = EncoderArena(frozen[0], trainable=False) if frozen else None
        assert self.arena_k.total == self.arena_q.total and self.arena_k.param_slices == self.arena_q.param_slices
        self.arena_k.copy_from(self.arena_q)
        self.momentum_encoder.bind(self.arena_k, self.arena_q)
        self._alpha = torch.full((1,), 1.0 / float(T), dtype=torch.float32, device=dev)
        self._labels = {}

    @staticmethod
    def _build_projector_and_predictor_mlps(owner, dim, mlp_dim):
        hidden_dim = owner.base_encoder.head.weight.shape[0]
        del owner.base_encoder.head                     # remove original fc layer
        owner.base_encoder.head = build_mlp(3, hidden_dim, mlp_dim, dim)      # projector
        owner.predictor = build_mlp(2, dim, mlp_dim, dim)                     # predictor

    # -- state plumbing --------------------------------------------------------------------------------------
    def _arenas(self):
        return [a for a in (self.arena_q, self.arena_k, self.arena_pe) if a is not None]

    def sync_runtime_state(self):
        for a in self._arenas():
            a.refresh()
        self.momentum_encoder.sync_steps()

    def load_state_dict(self, state_dict, strict=True):
        r = super().load_state_dict(state_dict, strict=strict)
        self.sync_runtime_state()
        return r

    # -- reference API ---------------------------------------------------------------------------------------
    def contrastive_loss(self, q, k):
        k = ops.l2norm_fwd(k.contiguous(), 1e-12)[0]
        k = concat_all_gather(k)
        logits = _KeyLogitsFn.apply(q, k, self._alpha)
        N = logits.shape[0]
        rank = dist.get_rank() if collectives_active() else 0
        key = (N, rank)
        if key not in self._labels:
            self._labels[key] = torch.arange(N, dtype=torch.int64, device=logits.device) + N * rank
        return _RowCEFn.apply(logits, self._labels[key]) * (2 * self.T)

    def _encode(self, base, predictor, x1, x2):
        N = x1.shape[0]
        feats = base.forward_features(torch.cat([x1, x2], dim=0))        # the trunk sees both views at once
        return predictor(base.head(feats[:N])), predictor(base.head(feats[N:]))

    def forward(self, inputs):
        assert isinstance(inputs, (list, tuple))
        x1, x2 = inputs[0], inputs[1]
        self.arena_q.refresh()                       # compute-dtype operands from the fp32 masters (after AdamW)
        q1, q2 = self._encode(self.base_encoder, self.predictor, x1, x2)
        with torch.no_grad():                        # no gradient
            self.momentum_encoder.update_parameters()
            mom = self.momentum_encoder.model
            k1, k2 = self._encode(mom[0], mom[1], x1, x2)
        return self.contrastive_loss(q1, k2) + self.contrastive_loss(q2, k1)

    def load_pretrained(self, path, rank=0, finetune=False):
        if not os.path.exists(path + '.pdparams'):
            raise ValueError('Model pretrain path {} does not exists.'.format(path))
        load_lenient(self, load_pickle(path + '.pdparams'), what='pretrained model')
        self.sync_runtime_state()

    def save(self, path, local_rank=0, rank=0):
        """<path>.pdparams = the whole state; <path>_base_encoder.pdparams = the backbone without the projector,
        prefix removed (mocov3.py:247-262)."""
        if rank != 0:
            return
        os.makedirs(os.path.dirname(path) or '.', exist_ok=True)
        sd = to_numpy(dict(self.state_dict()))
        with open(path + '.pdparams', 'wb') as f:
            pickle.dump(sd, f, protocol=2)
        enc = {k[len('base_encoder.'):]: v for k, v in sd.items()
               if k.startswith('base_encoder') and not k.startswith('base_encoder.head')}
        with open(path + '_base_encoder.pdparams', 'wb') as f:
            pickle.dump(enc, f, protocol=2)


def mocov3_vit_base(**kwargs):
    return MoCoV3ViT(patch_size=16, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4, qkv_bias=True,
                     norm_layer=partial(hnn.LayerNorm, epsilon=1e-6), **kwargs)


def mocov3_vit_base_linearprobe(**kwargs):
    return MoCoV3LinearProbe(patch_size=16, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4, qkv_bias=True,
                             norm_layer=partial(hnn.LayerNorm, epsilon=1e-6), **kwargs)


def mocov3_vit_base_pretrain(**kwargs):
    base_encoder = partial(mocov3_vit_base, stop_grad_conv1=True)
    return MoCoV3Pretrain(base_encoder=base_encoder, dim=256, mlp_dim=4096, T=0.2, base_momentum=0.99, **kwargs)
