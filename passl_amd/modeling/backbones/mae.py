"""MAE (masked autoencoder, ViT encoder + light decoder) on the MI355X HIP path.

Constructor, registry name, sub-layer / state_dict names and the ``forward(imgs, mask_ratio)``
-> ``(loss, pred, mask)`` contract are the reference's ``MAE`` (passl_v110/modeling/backbones/
mae.py:318-564, twin of passl/models/mae.py:37-290); Mlp / Attention / Block follow :61-189.
Execution: tokens are 2-D rows [B*T, C] in the compute dtype; every Linear is the implicit-GEMM
kernel (bias, residual add in the epilogue; fp32 output for the pixel prediction), LayerNorm / GELU /
attention / token gather-unshuffle / patchify / masked-patch loss are HIP kernels
(csrc/vit.hip, csrc/attention.hip).  ``pos_embed`` / ``decoder_pos_embed`` are fixed sin-cos tables
(reference: parameters with stop_gradient=True) kept as buffers under the same state_dict keys.
The per-sample noise of random_masking comes from ``torch.rand`` on the device (``noise=`` lets
tests inject the reference's draw); the argsort pair is replaced by a rank kernel."""
import math
from functools import partial

import torch
import torch.nn as tnn
from torch.autograd import Function

from ...hip import config, nn, ops, plan as P
from ...modules.get_sincos_pe import get_2d_sincos_pos_embed
from .builder import BACKBONES


@torch.no_grad()
def xavier_uniform_(w, fan_in, fan_out):
    a = math.sqrt(6.0 / (fan_in + fan_out))
    w.copy_((torch.rand(w.shape) * 2 - 1) * a)


@torch.no_grad()
def trunc_normal_(w, std=0.02):
    # paddle TruncatedNormal(std): N(0, std) re-sampled (not wrapped) into [-2 std, 2 std]
    t = torch.empty(w.shape)
    torch.nn.init.trunc_normal_(t, mean=0.0, std=std, a=-2.0 * std, b=2.0 * std)
    w.copy_(t)


class Identity(nn.Layer):
    def forward(self, x):
        return x


class Mlp(nn.Layer):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.):
        super().__init__()
        assert drop == 0., 'dropout is not used by the MAE pre-training recipe'
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features, out_features)

    def forward(self, x, residual=None):
        return self.fc2(self.act(self.fc1(x)), residual=residual)


class _PatchProj(nn.Layer):
    """The 16x16/stride-16 patch-embedding convolution as a GEMM over patchified rows.  ``weight`` is
    logically [embed_dim, in_chans, p, p] (reference layout) and physically [embed_dim][p][p][in_chans]
    = the K-order the patchify kernel writes."""
    krsc_weight = True
    no_dgrad = True           # the image needs no gradient

    def __init__(self, in_chans, embed_dim, patch, bias=True):
        super().__init__()
        dev = config.get_device()
        self.patch, self.in_chans, self.out_features = patch, in_chans, embed_dim
        self.in_features = in_chans * patch * patch
        self.geom = P.ConvGeom(self.in_features, embed_dim, 1, 1, 0)
        self.weight = tnn.Parameter(torch.empty(embed_dim, in_chans, patch, patch, device=dev))
        self.bias = tnn.Parameter(torch.zeros(embed_dim, device=dev)) if bias else None
        self._rt = None
        self._plans = {}

    _plan = nn.Linear._plan

    def forward(self, rows):
        return nn._LinearFn.apply(rows, self.weight, self.bias, self, False, False, None)


class PatchEmbed(nn.Layer):
    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768, norm_layer=None, flatten=True):
        super().__init__()
        self.img_size = (img_size, img_size)
        self.patch_size = (patch_size, patch_size)
        self.grid_size = (img_size // patch_size, img_size // patch_size)
        self.num_patches = self.grid_size[0] * self.grid_size[1]
        self.proj = _PatchProj(in_chans, embed_dim, patch_size)
        self.norm = norm_layer(embed_dim) if norm_layer else Identity()

    def forward(self, x):
        B, C, H, W = x.shape
        assert H == self.img_size[0], f"Input image height ({H}) doesn't match model ({self.img_size[0]})."
        assert W == self.img_size[1], f"Input image width ({W}) doesn't match model ({self.img_size[1]})."
        dtype = nn._need_rt(self.proj).arena.dtype
        rows = ops.patchify(x.contiguous().float(), self.patch_size[0], dtype)
        return self.norm(self.proj(rows))                    # [B*L, embed_dim]


class Attention(nn.Layer):
    def __init__(self, dim, num_heads=8, qkv_bias=False, attn_drop=0., proj_drop=0.):
        super().__init__()
        assert attn_drop == 0. and proj_drop == 0.
        self.num_heads = num_heads
        self.head_dim = dim // num_heads
        self.scale = self.head_dim ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias_attr=None if qkv_bias else False)
        self.proj = nn.Linear(dim, dim)

    def forward(self, x, B, T, residual=None):
        a = nn.attention(self.qkv(x), B, T, self.num_heads, self.head_dim, self.scale)
        return self.proj(a, residual=residual)


class Block(nn.Layer):
    def __init__(self, dim, num_heads, mlp_ratio=4., qkv_bias=False, drop=0., attn_drop=0., drop_path=0.,
                 act_layer=nn.GELU, norm_layer=nn.LayerNorm):
        super().__init__()
        assert drop_path == 0., 'stochastic depth is not used by the MAE pre-training recipe'
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, attn_drop=attn_drop, proj_drop=drop)
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=drop)

    def forward(self, x, B, T):
        # x + attn(norm1(x)) and x + mlp(norm2(x)): the add runs in the epilogue of proj / fc2, the
        # fork's gradient add inside the LayerNorm backward kernel (nn.LayerNorm.fork)
        h, xr = self.norm1.fork(x)
        x = self.attn(h, B, T, residual=xr)
        h, xr = self.norm2.fork(x)
        return self.mlp(h, residual=xr)


class _GatherFn(Function):
    @staticmethod
    def forward(ctx, x, cls_token, pos, ids_keep, ids_restore, B, L):
        ctx.save_for_backward(ids_restore)
        ctx.cls, ctx.dims = cls_token, (B, L, ids_keep.shape[1])
        nn.param_expect_grad(cls_token)
        return ops.mae_gather(x, cls_token.detach().view(-1), pos.view(-1, pos.shape[-1]), ids_keep, B, L)

    @staticmethod
    def backward(ctx, dout):
        (ids_restore,) = ctx.saved_tensors
        B, L, K = ctx.dims
        cls = ctx.cls
        if cls.grad is None:
            cls.grad = torch.zeros_like(cls)
        dx = ops.mae_gather_bwd(dout.contiguous(), ids_restore, cls.grad, B, L, K)
        nn.param_grad_ready(cls)
        return dx, None, None, None, None, None, None


class _UnshuffleFn(Function):
    @staticmethod
    def forward(ctx, x, mask_token, pos, ids_keep, ids_restore, B):
        ctx.save_for_backward(ids_keep, ids_restore)
        ctx.tok, ctx.B = mask_token, B
        nn.param_expect_grad(mask_token)
        return ops.mae_unshuffle(x, mask_token.detach().view(-1), pos.view(-1, pos.shape[-1]), ids_restore,
                                 B, ids_keep.shape[1])

    @staticmethod
    def backward(ctx, dout):
        ids_keep, ids_restore = ctx.saved_tensors
        tok = ctx.tok
        if tok.grad is None:
            tok.grad = torch.zeros_like(tok)
        dx = ops.mae_unshuffle_bwd(dout.contiguous(), ids_keep, ids_restore, tok.grad, ctx.B)
        nn.param_grad_ready(tok)
        return dx, None, None, None, None, None


from ...loss.mae import masked_patch_loss      # the fused loss lives in passl.loss.mae


@BACKBONES.register()
class MAE(nn.Layer):
    """Masked Autoencoder with VisionTransformer backbone."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=1024, depth=24, num_heads=16,
                 decoder_embed_dim=512, decoder_depth=8, decoder_num_heads=16, mlp_ratio=4.,
                 norm_layer=partial(nn.LayerNorm, epsilon=1e-6), norm_pix_loss=False):
        super().__init__()
        dev = config.get_device()
        self.patch_embed = PatchEmbed(img_size, patch_size, in_chans, embed_dim)
        num_patches = self.patch_embed.num_patches
        self.cls_token = tnn.Parameter(torch.zeros(1, 1, embed_dim, device=dev))
        self.register_buffer('pos_embed', torch.zeros(1, num_patches + 1, embed_dim, device=dev))
        self.blocks = tnn.ModuleList([Block(embed_dim, num_heads, mlp_ratio, qkv_bias=True, norm_layer=norm_layer)
                                      for _ in range(depth)])
        self.norm = norm_layer(embed_dim)
        self.decoder_embed = nn.Linear(embed_dim, decoder_embed_dim)
        self.mask_token = tnn.Parameter(torch.zeros(1, 1, decoder_embed_dim, device=dev))
        self.register_buffer('decoder_pos_embed', torch.zeros(1, num_patches + 1, decoder_embed_dim, device=dev))
        self.decoder_blocks = tnn.ModuleList([Block(decoder_embed_dim, decoder_num_heads, mlp_ratio, qkv_bias=True,
                                                    norm_layer=norm_layer) for _ in range(decoder_depth)])
        self.decoder_norm = norm_layer(decoder_embed_dim)
        self.decoder_pred = nn.Linear(decoder_embed_dim, patch_size ** 2 * in_chans)
        self.norm_pix_loss = norm_pix_loss
        self.in_chans = in_chans
        self.initialize_weights()

    @torch.no_grad()
    def initialize_weights(self):
        g = int(self.patch_embed.num_patches ** .5)
        self.pos_embed.copy_(torch.from_numpy(
            get_2d_sincos_pos_embed(self.pos_embed.shape[-1], g, cls_token=True)).float().unsqueeze(0))
        self.decoder_pos_embed.copy_(torch.from_numpy(
            get_2d_sincos_pos_embed(self.decoder_pos_embed.shape[-1], g, cls_token=True)).float().unsqueeze(0))
        trunc_normal_(self.cls_token, std=0.02)              # create_parameter(default_initializer=trunc_normal_)
        w = self.patch_embed.proj.weight                      # xavier_uniform_ on the [D, C*p*p] view
        xavier_uniform_(w, w.shape[1] * w.shape[2] * w.shape[3], w.shape[0])
        for m in self.modules():
            if isinstance(m, nn.Linear):
                xavier_uniform_(m.weight, m.weight.shape[0], m.weight.shape[1])
                if m.bias is not None:
                    m.bias.zero_()
            elif isinstance(m, nn.LayerNorm):
                m.bias.zero_()
                m.weight.fill_(1.0)

    # -- reference helpers kept for API parity (host-side, not on the hot path) ------------------
    def patchify(self, imgs):
        p = self.patch_embed.patch_size[0]
        h = w = imgs.shape[2] // p
        x = imgs.reshape(imgs.shape[0], self.in_chans, h, p, w, p)
        return torch.einsum('nchpwq->nhwpqc', x).reshape(imgs.shape[0], h * w, p ** 2 * self.in_chans)

    def unpatchify(self, x):
        p = self.patch_embed.patch_size[0]
        h = w = int(x.shape[1] ** .5)
        x = x.reshape(x.shape[0], h, w, p, p, self.in_chans)
        return torch.einsum('nhwpqc->nchpwq', x).reshape(x.shape[0], self.in_chans, h * p, h * p)

    def random_masking_ids(self, B, L, mask_ratio, noise=None):
        len_keep = int(L * (1 - mask_ratio))
        if noise is None:
            # paddle.rand([N, L]).  torch.rand IS empty + uniform_ (same generator stream); written this way the draw is a
            # host call that stays live when the step is replayed from a native plan (hip/replay.py) — a fresh mask
            # at every replay, in the recorded buffer
            from ...hip.replay import host_call
            noise = torch.empty(B, L, device=self.cls_token.device)
            host_call(lambda: noise.uniform_())
        ids_keep, ids_restore, mask = ops.mae_mask(noise.contiguous().float(), len_keep)
        return ids_keep, ids_restore, mask, len_keep

    def forward_encoder(self, imgs, mask_ratio, noise=None):
        B = imgs.shape[0]
        L = self.patch_embed.num_patches
        x = self.patch_embed(imgs)                                       # [B*L, D]
        ids_keep, ids_restore, mask, K = self.random_masking_ids(B, L, mask_ratio, noise)
        x = _GatherFn.apply(x, self.cls_token, self.pos_embed, ids_keep, ids_restore, B, L)
        for blk in self.blocks:
            x = blk(x, B, K + 1)
        return self.norm(x), mask, (ids_keep, ids_restore)

    def forward_decoder(self, x, ids, B):
        ids_keep, ids_restore = ids
        L = ids_restore.shape[1]
        x = self.decoder_embed(x)
        x = _UnshuffleFn.apply(x, self.mask_token, self.decoder_pos_embed, ids_keep, ids_restore, B)
        for blk in self.decoder_blocks:
            x = blk(x, B, L + 1)
        x = self.decoder_norm(x)
        return self.decoder_pred(x, out_f32=True)                        # [B*(L+1), p*p*3] fp32, cls rows included

    def forward_loss(self, imgs, pred_rows, mask):
        denom = float(mask.shape[0] * (mask.shape[1] - int(self._len_keep)))
        return masked_patch_loss(pred_rows, imgs, mask, self.patch_embed.patch_size[0], self.norm_pix_loss, denom)

    def forward(self, imgs, mask_ratio=0.75, noise=None):
        B = imgs.shape[0]
        L = self.patch_embed.num_patches
        self._len_keep = int(L * (1 - mask_ratio))
        latent, mask, ids = self.forward_encoder(imgs, mask_ratio, noise)
        pred_rows = self.forward_decoder(latent, ids, B)
        loss = self.forward_loss(imgs, pred_rows, mask)
        pred = pred_rows.view(B, L + 1, -1)[:, 1:, :]                    # remove cls token
        return loss, pred, mask


# ============================================================ fine-tuning trunk (configs/mae/mae_vit_b_finetune.yaml)
class _PatchMeanFn(Function):
    """``x[:, 1:, :].mean(axis=1)`` over token rows [B*(L+1), D] (mae.py:308: global average pool of the patch tokens,
    class token excluded): the NHWC average-pool kernel over all L+1 tokens, minus the class row."""

    @staticmethod
    def forward(ctx, x, cls_rows, B, L):
        D = x.shape[-1]
        ctx.dims = (B, L, D)
        avg = ops.avgpool_fwd(x.view(B, L + 1, 1, D))
        cls = ops.gather_rows(x, cls_rows)
        return ((avg.float() * (L + 1) - cls.float()) / L).to(x.dtype)

    @staticmethod
    def backward(ctx, dout):
        B, L, D = ctx.dims
        # every patch token receives dout / L: the pool's backward hands out dy / (L + 1)
        dx = ops.avgpool_bwd((dout.float() * ((L + 1.0) / L)).to(dout.dtype).contiguous(), L + 1, 1)
        dx = dx.view(B, L + 1, D)
        dx[:, 0].zero_()
        return dx.view(B * (L + 1), D), None, None, None


class VisionTransformer(nn.Layer):
    """The fine-tuning ViT of the v110 tree (passl_v110/modeling/backbones/mae.py:190-277): learnable ``cls_token`` /
    ``pos_embed`` (trunc-normal 0.02), pre-norm blocks, final LayerNorm, output = the class token's row.  Same kernels as
    the pre-training encoder above; dropout / stochastic depth are not built (0 in the yaml's defaults)."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, num_classes=1000, embed_dim=768, depth=12,
                 num_heads=12, mlp_ratio=4., qkv_bias=True, drop_rate=0., attn_drop_rate=0., drop_path_rate=0.,
                 embed_layer=PatchEmbed, norm_layer=None, act_layer=None, weight_init=''):
        super().__init__()
        if drop_rate or attn_drop_rate or drop_path_rate:
            raise NotImplementedError('dropout / stochastic depth are not built on the HIP path')
        dev = config.get_device()
        self.num_classes = num_classes
        self.num_features = self.embed_dim = embed_dim
        norm_layer = norm_layer or partial(nn.LayerNorm, epsilon=1e-6)
        act_layer = act_layer or nn.GELU
        self.patch_embed = embed_layer(img_size=img_size, patch_size=patch_size, in_chans=in_chans, embed_dim=embed_dim)
        num_patches = self.patch_embed.num_patches
        if embed_dim % num_heads or embed_dim // num_heads not in ops.ATTENTION_HEAD_DIMS or \
                num_patches + 1 > ops.ATTENTION_MAX_TOKENS:
            raise NotImplementedError('%d tokens x head dimension %s is outside the HIP attention kernels'
                                      % (num_patches + 1, embed_dim / float(num_heads)))
        self.cls_token = tnn.Parameter(torch.zeros(1, 1, embed_dim, device=dev))
        self.pos_embed = tnn.Parameter(torch.zeros(1, num_patches + 1, embed_dim, device=dev))
        self.blocks = tnn.Sequential(*[Block(embed_dim, num_heads, mlp_ratio, qkv_bias=qkv_bias, norm_layer=norm_layer,
                                             act_layer=act_layer) for _ in range(depth)])
        self.norm = norm_layer(embed_dim)
        self._ids = {}
        trunc_normal_(self.cls_token, std=0.02)
        trunc_normal_(self.pos_embed, std=0.02)
        with torch.no_grad():
            # self.apply(self._init_weights): Linear trunc_normal(.02) / zero bias, LayerNorm (1, 0); the patch
            # convolution keeps nn.Conv2D's default [Paddle-semantics]: Normal(0, sqrt(2 / fan_in)), zero bias
            w = self.patch_embed.proj.weight
            w.copy_(torch.randn(w.shape) * math.sqrt(2.0 / (w.shape[1] * w.shape[2] * w.shape[3])))
            for m in self.modules():
                if isinstance(m, nn.Linear):
                    trunc_normal_(m.weight, std=0.02)
                    if m.bias is not None:
                        m.bias.zero_()

    def _token_ids(self, B, L, device):
        key = (B, L)
        if key not in self._ids:
            self._ids[key] = (torch.arange(L, dtype=torch.int32, device=device).repeat(B, 1).contiguous(),
                              (torch.arange(B, dtype=torch.int32, device=device) * (L + 1)).contiguous())
        return self._ids[key]

    def _tokens(self, x):
        from .vision_transformer import _ClsPosFn
        B = x.shape[0]
        L = self.patch_embed.num_patches
        x = self.patch_embed(x)                                           # [B*L, D]
        ids, cls_rows = self._token_ids(B, L, x.device)
        x = _ClsPosFn.apply(x, self.cls_token, self.pos_embed, ids, B, L)     # concat(cls, x) + pos_embed
        for blk in self.blocks:
            x = blk(x, B, L + 1)
        return x, cls_rows, B, L

    def forward_features(self, x):
        x, cls_rows, _B, _L = self._tokens(x)
        return self.norm(nn.gather_rows(x, cls_rows))                     # norm(x)[:, 0]: LayerNorm is per token

    def forward(self, x):
        return self.forward_features(x)


@BACKBONES.register()
class MAE_ViT(VisionTransformer):
    """Vision Transformer with support for global average pooling — passl_v110/modeling/backbones/mae.py:279-314:
    with ``global_pool`` the final ``norm`` is replaced by ``fc_norm`` over the mean of the patch tokens."""

    def __init__(self, global_pool=True, **kwargs):
        super().__init__(**kwargs)
        self.global_pool = global_pool
        if self.global_pool:
            self.fc_norm = nn.LayerNorm(kwargs['embed_dim'], epsilon=1e-6)
            del self.norm                                                  # remove the original norm

    def forward_features(self, x):
        x, cls_rows, B, L = self._tokens(x)
        if self.global_pool:
            return self.fc_norm(_PatchMeanFn.apply(x, cls_rows, B, L))
        return self.norm(nn.gather_rows(x, cls_rows))
