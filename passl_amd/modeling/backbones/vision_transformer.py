"""ViT building blocks of the CLIP path on the MI355X HIP kernels.

Constructor arguments, registry name and sub-layer / state_dict names follow
passl_v110/modeling/backbones/vision_transformer.py: Mlp :69-93, Attention :96-137 (additive
attention mask before the softmax), Block :140-189 (pre-norm, QuickGELU, LayerNorm eps 1e-5),
Transformer :192-225, PatchEmbed :228-264, VisionTransformer :267-366 (patch conv, class_embedding
concat, + positional_embedding, norm_pre, blocks, ``norm_post(x[:, 0]) @ proj``).
Execution: tokens are 2-D rows [B*T, C] in the compute dtype; Linears are the implicit-GEMM kernel
(bias / residual add in the epilogue), LayerNorm / QuickGELU / attention / token plumbing are HIP
kernels (csrc/vit.hip, csrc/attention.hip, csrc/clip.hip).  The only attention mask the reference
builds is CLIP's causal ``triu(-inf, 1)`` (clip.py:284-286): ``attn_mask`` is accepted as the
string ``'causal'`` or a tensor equal to that matrix and becomes the kernels' causal flag; any other
mask raises.  The raw matrix parameters ``proj`` (and CLIP's ``text_projection``) keep their
state_dict key but live in a bias-free Linear so that they run on the GEMM kernels
(``named_parameters`` shows them as ``<name>.weight``)."""
import math

import numpy as np
import torch
import torch.nn as tnn
from torch.autograd import Function

from ...hip import config, nn, ops
from .builder import BACKBONES
from .mae import _PatchProj, trunc_normal_


def to_2tuple(x):
    return tuple([x] * 2)


class Identity(nn.Layer):
    def forward(self, x):
        return x


QuickGELU = nn.QuickGELU
_NORMS = {'nn.LayerNorm': nn.LayerNorm}


def _norm_layer(norm_layer):
    """The reference evaluates the string ``norm_layer`` ("nn.LayerNorm")."""
    if isinstance(norm_layer, str):
        if norm_layer not in _NORMS:
            raise NotImplementedError('norm_layer %r (supported: %s)' % (norm_layer, sorted(_NORMS)))
        return _NORMS[norm_layer]
    return norm_layer


def _is_causal(attn_mask):
    if attn_mask is None:
        return False
    if isinstance(attn_mask, str):
        if attn_mask != 'causal':
            raise NotImplementedError('attn_mask %r' % attn_mask)
        return True
    m = torch.as_tensor(attn_mask).float().cpu()
    T = m.shape[-1]
    ref = torch.triu(torch.full((T, T), -math.inf), 1)
    if m.shape != ref.shape or not torch.equal(m, ref):
        raise NotImplementedError('only the causal triu(-inf, 1) attention mask is supported by the HIP '
                                  'attention kernel')
    return True


def alias_matrix_param(parent, attr):
    """state_dict key '<attr>' <-> the bias-free Linear parent.<attr>.weight."""
    def save_hook(module, sd, prefix, local_metadata):
        k = prefix + attr + '.weight'
        if k in sd:
            sd[prefix + attr] = sd.pop(k)

    def load_hook(sd, prefix, *args):
        k = prefix + attr
        if k in sd:
            sd[prefix + attr + '.weight'] = sd.pop(k)
    parent._register_state_dict_hook(save_hook)
    parent._register_load_state_dict_pre_hook(load_hook)


class Mlp(nn.Layer):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.0):
        super().__init__()
        assert drop == 0.0, 'dropout is not used by the CLIP recipe'
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features, out_features)

    def forward(self, x, residual=None):
        return self.fc2(self.act(self.fc1(x)), residual=residual)


class Attention(nn.Layer):
    def __init__(self, dim, num_heads=8, qkv_bias=True, qk_scale=None, attn_mask=None, attn_drop=0.0,
                 proj_drop=0.0):
        super().__init__()
        assert attn_drop == 0.0 and proj_drop == 0.0
        self.num_heads = num_heads
        self.head_dim = dim // num_heads
        self.scale = qk_scale or self.head_dim ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias_attr=None if qkv_bias else False)
        self.causal = _is_causal(attn_mask)
        self.proj = nn.Linear(dim, dim)

    def forward(self, x, B, T, residual=None):
        a = nn.attention(self.qkv(x), B, T, self.num_heads, self.head_dim, self.scale, causal=self.causal)
        return self.proj(a, residual=residual)


class Block(nn.Layer):
    def __init__(self, dim, num_heads, mlp_ratio=4.0, qkv_bias=False, qk_scale=None, drop=0.0, attn_mask=None,
                 attn_drop=0.0, drop_path=0.0, act_layer=QuickGELU, norm_layer='nn.LayerNorm', epsilon=1e-5):
        super().__init__()
        assert drop_path == 0.0, 'stochastic depth is not used by the CLIP recipe'
        norm = _norm_layer(norm_layer)
        self.norm1 = norm(dim, epsilon=epsilon)
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, qk_scale=qk_scale, attn_mask=attn_mask,
                              attn_drop=attn_drop, proj_drop=drop)
        self.drop_path = Identity()
        self.norm2 = norm(dim, epsilon=epsilon)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=drop)

    def forward(self, x, B, T):
        # x + attn(norm1(x)) and x + mlp(norm2(x)): the add runs in the epilogue of proj / fc2, the
        # fork's gradient add inside the LayerNorm backward kernel (nn.LayerNorm.fork)
        h, xr = self.norm1.fork(x)
        x = self.attn(h, B, T, residual=xr)
        h, xr = self.norm2.fork(x)
        return self.mlp(h, residual=xr)


class Transformer(nn.Layer):
    def __init__(self, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4, qkv_bias=True, qk_scale=None,
                 drop_rate=0.0, attn_mask=None, attn_drop_rate=0.0, drop_path_rate=0.0,
                 norm_layer='nn.LayerNorm', epsilon=1e-5, **args):
        super().__init__()
        self.embed_dim = embed_dim
        self.depth = depth
        dpr = np.linspace(0, drop_path_rate, depth)
        self.blocks = tnn.ModuleList([
            Block(dim=embed_dim, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_scale=qk_scale,
                  drop=drop_rate, attn_mask=attn_mask, attn_drop=attn_drop_rate, drop_path=float(dpr[i]),
                  norm_layer=norm_layer, epsilon=epsilon) for i in range(depth)])

    def forward(self, x, B, T):
        for blk in self.blocks:
            x = blk(x, B, T)
        return x


class PatchEmbed(nn.Layer):
    """Image to Patch Embedding: the p x p / stride-p convolution as a GEMM over patchified rows."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768, patch_bias=True):
        super().__init__()
        img_size, patch_size = to_2tuple(img_size), to_2tuple(patch_size)
        self.patches_resolution = [img_size[0] // patch_size[0], img_size[1] // patch_size[1]]
        self.img_size, self.patch_size = img_size, patch_size
        self.num_patches = self.patches_resolution[0] * self.patches_resolution[1]
        self.proj = _PatchProj(in_chans, embed_dim, patch_size[0], bias=patch_bias)

    def forward(self, x):
        B, C, H, W = x.shape
        assert H == self.img_size[0] and W == self.img_size[1], \
            "Input image size (%d*%d) doesn't match model (%d*%d)." % (H, W, self.img_size[0], self.img_size[1])
        dtype = nn._need_rt(self.proj).arena.dtype
        return self.proj(ops.patchify(x.contiguous().float(), self.patch_size[0], dtype))   # [B*L, D]


class _ClsPosFn(Function):
    """rows[b, 0] = class_embedding + pos[0];  rows[b, 1 + l] = x[b, l] + pos[1 + l]
    (vision_transformer.py:352-357: expand + concat + add) and its backward."""

    @staticmethod
    def forward(ctx, x, cls, pos, ids, B, L):
        ctx.save_for_backward(ids)
        ctx.params, ctx.dims = (cls, pos), (B, L)
        nn.param_expect_grad(cls, pos)
        return ops.mae_gather(x, cls.detach().view(-1), pos.detach().view(-1, pos.shape[-1]), ids, B, L)

    @staticmethod
    def backward(ctx, dout):
        (ids,) = ctx.saved_tensors
        cls, pos = ctx.params
        B, L = ctx.dims
        for p in (cls, pos):
            if p.grad is None:
                p.grad = torch.zeros_like(p)
        dout = dout.contiguous()
        dx = ops.mae_gather_bwd(dout, ids, cls.grad, B, L, L)              # dx rows + dcls += sum_b dout[b, 0]
        ops.colsum_into(dout.view(B, -1), pos.grad.view(-1), accumulate=True)   # dpos[t] += sum_b dout[b, t]
        nn.param_grad_ready(cls, pos)
        return dx, None, None, None, None, None


@BACKBONES.register()
class VisionTransformer(nn.Layer):
    """Vision Transformer with support for patch input (the CLIP image tower)."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, class_dim=0, width=768, out_dim=512, depth=12,
                 num_heads=12, mlp_ratio=4, qkv_bias=True, qk_scale=None, drop_rate=0.0, attn_drop_rate=0.0,
                 drop_path_rate=0.0, norm_layer='nn.LayerNorm', pre_norm=False, proj=False, output_cls_token=True,
                 patch_bias=True, epsilon=1e-5, **args):
        super().__init__()
        assert drop_rate == 0.0
        dev = config.get_device()
        self.class_dim = class_dim
        self.num_features = self.width = width
        self.patch_embed = PatchEmbed(img_size=img_size, patch_size=patch_size, in_chans=in_chans, embed_dim=width,
                                      patch_bias=patch_bias)
        num_patches = self.patch_embed.num_patches
        scale = width ** -0.5
        self.class_embedding = tnn.Parameter(torch.zeros(1, 1, width, device=dev))
        self.positional_embedding = tnn.Parameter(torch.zeros(1, num_patches + 1, width, device=dev))
        self.proj = None
        if proj:
            self.proj = nn.Linear(width, out_dim, bias_attr=False)
            alias_matrix_param(self, 'proj')
        self.output_cls_token = output_cls_token
        norm = _norm_layer(norm_layer)
        self.norm_pre = norm(width, epsilon=epsilon) if pre_norm else Identity()
        dpr = np.linspace(0, drop_path_rate, depth)
        self.blocks = tnn.ModuleList([
            Block(dim=width, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_scale=qk_scale,
                  drop=drop_rate, attn_drop=attn_drop_rate, drop_path=float(dpr[i]), norm_layer=norm_layer,
                  epsilon=epsilon) for i in range(depth)])
        self.norm_post = norm(width, epsilon=epsilon)
        self._ids = {}
        with torch.no_grad():
            # vision_transformer.py:331-342: every Linear trunc_normal(.02) / zero bias, LayerNorm (1, 0);
            # the patch conv keeps nn.Conv2D's default Normal(0, sqrt(2 / (k*k*in)))
            w = self.patch_embed.proj.weight
            w.copy_(torch.randn(w.shape) * math.sqrt(2.0 / (w.shape[1] * w.shape[2] * w.shape[3])))
            trunc_normal_(self.positional_embedding)
            trunc_normal_(self.class_embedding)
            if self.proj is not None:
                self.proj.weight.copy_(torch.randn(self.proj.weight.shape) * scale)
            for m in self.modules():
                if isinstance(m, nn.Linear) and m is not self.proj:
                    trunc_normal_(m.weight)
                    if m.bias is not None:
                        m.bias.zero_()

    def _identity_ids(self, B, L, device):
        key = (B, L)
        if key not in self._ids:
            self._ids[key] = torch.arange(L, dtype=torch.int32, device=device).repeat(B, 1).contiguous()
            self._ids[('cls', B, L)] = (torch.arange(B, dtype=torch.int32, device=device) * (L + 1)).contiguous()
        return self._ids[key], self._ids[('cls', B, L)]

    def forward_features(self, x):
        if self.proj is None:
            raise NotImplementedError('VisionTransformer without `proj` (dense patch-token outputs) is outside '
                                      'the CLIP pre-training path')
        B = x.shape[0]
        L = self.patch_embed.num_patches
        x = self.patch_embed(x)                                           # [B*L, width]
        ids, cls_rows = self._identity_ids(B, L, x.device)
        x = _ClsPosFn.apply(x, self.class_embedding, self.positional_embedding, ids, B, L)
        x = self.norm_pre(x)
        for blk in self.blocks:
            x = blk(x, B, L + 1)
        x = self.norm_post(nn.gather_rows(x, cls_rows))                   # norm_post(x[:, 0, :])
        return self.proj(x, out_f32=True)                                 # @ proj, fp32 features

    def forward(self, x):
        return self.forward_features(x)
