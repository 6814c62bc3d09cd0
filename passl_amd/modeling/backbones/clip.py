"""CLIP (ViT image tower + causal text transformer) on the MI355X HIP kernels.

Constructor, registry name, sub-layer / state_dict names and the ``forward(image, text, is_train)``
-> ``(image_logits, text_logits)`` contract are the reference's ``CLIP``
(passl_v110/modeling/backbones/clip.py:183-336; the ModifiedResNet image tower of the RN50 variants,
:33-164, is not built — the shipped config configs/clip/vit-b-32.yaml uses the ViT tower).
Execution: image tower = vision_transformer.VisionTransformer; text tower: token embedding gather +
positional add (HIP), causal attention blocks, EOT-row gather (argmax of the token ids, HIP),
ln_final, text_projection GEMM with fp32 output; logits: L2 normalisation, exp(logit_scale) * I T^T
by exact-fp32 MFMA, in-place clip of logit_scale (csrc/clip.hip).  ``text_logits`` is returned as
the transpose VIEW of ``image_logits`` (equal up to the rounding of exp(s)*x in the reference);
CLIPHead relies on that to run both cross-entropies over one matrix."""
import math
import os

import numpy as np
import torch
import torch.distributed as dist
import torch.nn as tnn
from torch.autograd import Function

from ...core.sync_utils import collectives_active
from ...hip import config, nn, ops, streams
from .builder import BACKBONES
from .vision_transformer import Transformer, VisionTransformer, alias_matrix_param


class LayerNorm(nn.LayerNorm):
    """The reference subclasses LayerNorm to compute in fp32 for fp16 inputs (clip.py:168-175); the
    HIP LayerNorm kernel always accumulates in fp32."""


QuickGELU = nn.QuickGELU


class Embedding(nn.Layer):
    """paddle.nn.Embedding(vocab, dim): ``weight`` [vocab, dim] fp32; lookup fused with the positional
    add in CLIP.encode_text."""

    def __init__(self, num_embeddings, embedding_dim):
        super().__init__()
        self.weight = tnn.Parameter(torch.zeros(num_embeddings, embedding_dim, device=config.get_device()))


class _EmbedFn(Function):
    @staticmethod
    def forward(ctx, text, table, pos, dtype):
        ctx.save_for_backward(text)
        ctx.params = (table, pos)
        nn.param_expect_grad(table, pos)
        return ops.embed_fwd(text, table.detach(), pos.detach(), dtype)

    @staticmethod
    def backward(ctx, dout):
        (text,) = ctx.saved_tensors
        table, pos = ctx.params
        for p in (table, pos):
            if p.grad is None:
                p.grad = torch.zeros_like(p)
        ops.embed_bwd(text, dout.contiguous(), table.grad, pos.grad)
        nn.param_grad_ready(table, pos)
        return None, None, None, None


class _LogitsFn(Function):
    @staticmethod
    def forward(ctx, img, txt, logit_scale):
        logits, ws = ops.clip_logits_fwd(img.contiguous(), txt.contiguous(), logit_scale.detach())
        ctx.save_for_backward(logits, ws)
        ctx.scale, ctx.D = logit_scale, img.shape[1]
        nn.param_expect_grad(logit_scale)
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        logits, ws = ctx.saved_tensors
        s = ctx.scale
        if s.grad is None:
            s.grad = torch.zeros_like(s)
        dimg, dtxt = ops.clip_logits_bwd(dlogits.contiguous(), logits, ws, ctx.D, s.grad)
        nn.param_grad_ready(s)
        return dimg, dtxt, None


class _OffMainBoundary(Function):
    """Identity at the exit of a sub-graph that runs on another stream.  Its backward is the first node of that
    sub-graph's backward pass and runs on that stream (autograd's stream affinity); autograd orders it behind the
    producer of its incoming gradient with an event of its own, which a recording step plan cannot see — the plan gets the
    conservative equivalent (hip/streams.py:autograd_node_entry)."""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        streams.autograd_node_entry(g.device)
        return g


def _gather_rows_all(t):
    """all_gather + concat along dim 0 (identity without an active process group)."""
    if not collectives_active():
        return t
    out = torch.empty((dist.get_world_size() * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    src = t.contiguous()
    # (a live call also when the step is replayed from a native plan: the plan is cut here, hip/replay.py)
    from ...hip.replay import host_call
    host_call(lambda: dist.all_gather_into_tensor(out, src))
    return out


def _reduce_scatter_rows(t, rows):
    """Sum over ranks of t [W*rows, D], this rank's [rows, D] slice (identity without a process group)."""
    if not collectives_active():
        return t
    out = torch.empty((rows,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    src = t.contiguous()
    from ...hip.replay import host_call
    host_call(lambda: dist.reduce_scatter_tensor(out, src))
    return out


class _CrossRankLogitsFn(Function):
    """CLIP.forward's logits against the features of EVERY rank (BASELINE configs[4]: cross-GPU InfoNCE;
    the gather pattern of passl/models/mocov3.py:187-198 applied to both modalities):

        image_logits = exp(s) * I^_local . T^_all^T   [B, W*B]      text_logits = exp(s) * T^_local . I^_all^T

    with I^, T^ the L2-normalised features (no epsilon, clip.py:325-326) and s clipped in place after use
    (clip.py:309-311).  Row i's positive is column B*rank + i.  Backward: the local rows' gradient plus the
    reduce-scattered gradient of the gathered copies (every rank's text_logits contain this rank's image
    features as columns, and vice versa).  World size 1 gives (L, L^T) — the reference's two matrices."""

    @staticmethod
    def forward(ctx, img, txt, logit_scale):
        B, D = img.shape
        img_n, img_norm = ops.l2norm_fwd(img.contiguous(), 0.0)
        txt_n, txt_norm = ops.l2norm_fwd(txt.contiguous(), 0.0)
        alpha = ops.clip_scale(logit_scale.detach())
        img_all, txt_all = _gather_rows_all(img_n), _gather_rows_all(txt_n)
        li = ops.gemm_f32_nt(img_n, txt_all, alpha)
        lt = ops.gemm_f32_nt(txt_n, img_all, alpha)
        ctx.save_for_backward(img_n, img_norm, txt_n, txt_norm, img_all, txt_all, alpha, li, lt)
        ctx.scale = logit_scale
        nn.param_expect_grad(logit_scale)
        return li, lt

    @staticmethod
    def backward(ctx, dli, dlt):
        img_n, img_norm, txt_n, txt_norm, img_all, txt_all, alpha, li, lt = ctx.saved_tensors
        B = img_n.shape[0]
        dli, dlt = dli.contiguous(), dlt.contiguous()
        s = ctx.scale
        if s.grad is None:
            s.grad = torch.zeros_like(s)
        ops.dot_acc(dli, li, s.grad)                       # d loss / d s = sum dL o L for both matrices
        ops.dot_acc(dlt, lt, s.grad)
        nn.param_grad_ready(s)
        # row role (local rows) + column role (this rank's features inside every rank's other matrix)
        # (library adds: no framework launch inside a step that a native plan replays)
        dimg = ops.add_into(ops.gemm_f32_gx(dli, txt_all, alpha),
                            _reduce_scatter_rows(ops.gemm_f32_gx(dlt, txt_n, alpha, trans=True), B))
        dtxt = ops.add_into(ops.gemm_f32_gx(dlt, img_all, alpha),
                            _reduce_scatter_rows(ops.gemm_f32_gx(dli, img_n, alpha, trans=True), B))
        return (ops.l2norm_bwd(dimg, img_n, img_norm, torch.float32),
                ops.l2norm_bwd(dtxt, txt_n, txt_norm, torch.float32), None)


@BACKBONES.register()
class CLIP(nn.Layer):
    def __init__(self, embed_dim,
                 # vision
                 image_resolution, vision_layers, vision_width, vision_patch_size, pre_norm, proj, patch_bias,
                 # text
                 context_length, vocab_size, transformer_width, transformer_heads, transformer_layers, qkv_bias):
        super().__init__()
        dev = config.get_device()
        self.context_length = context_length
        if isinstance(vision_layers, (tuple, list)):
            raise NotImplementedError('the ModifiedResNet image tower (CLIP RN50 variants) is not built; '
                                      'configs/clip/vit-b-32.yaml uses the ViT tower')
        vision_heads = vision_width // 64
        self.visual = VisionTransformer(img_size=image_resolution, patch_size=vision_patch_size, width=vision_width,
                                        out_dim=embed_dim, depth=vision_layers, num_heads=vision_heads,
                                        pre_norm=pre_norm, proj=proj, patch_bias=patch_bias)
        # NB: like the reference, `qkv_bias` is not forwarded — both towers use the blocks' default (True)
        self.transformer = Transformer(embed_dim=transformer_width, depth=transformer_layers,
                                       num_heads=transformer_heads, attn_mask='causal')
        self.vocab_size = vocab_size
        self.token_embedding = Embedding(vocab_size, transformer_width)
        self.positional_embedding = tnn.Parameter(torch.zeros(context_length, transformer_width, device=dev))
        self.ln_final = LayerNorm(transformer_width)
        self.text_projection = nn.Linear(transformer_width, embed_dim, bias_attr=False)
        alias_matrix_param(self, 'text_projection')
        self.logit_scale = tnn.Parameter(torch.full((1,), float(np.log(1 / 0.07)), device=dev))
        self.initialize_parameters()

    def build_attention_mask(self, length):
        """The additive mask the reference hands to its Transformer (clip.py:284-286); the HIP attention
        kernels implement it as their causal flag."""
        return torch.triu(torch.full((length, length), -math.inf), 1)

    @torch.no_grad()
    def initialize_parameters(self):
        """clip.py:251-282 (the text blocks' proj std is width^-0.5 * (2*depth), as written there)."""
        def normal_(p, std):
            p.copy_(torch.randn(p.shape) * std)
        normal_(self.token_embedding.weight, 0.02)
        normal_(self.positional_embedding, 0.01)
        w = self.transformer.embed_dim
        proj_std = (w ** -0.5) * (2 * self.transformer.depth)
        attn_std = w ** -0.5
        fc_std = (2 * w) ** -0.5
        for block in self.transformer.blocks:
            for lin, std in ((block.attn.proj, proj_std), (block.attn.qkv, attn_std), (block.mlp.fc1, fc_std),
                             (block.mlp.fc2, proj_std)):
                normal_(lin.weight, std)
                lin.bias.zero_()
        normal_(self.text_projection.weight, w ** -0.5)

    @property
    def dtype(self):
        return nn._need_rt(self.visual.patch_embed.proj).arena.dtype

    def encode_image(self, image):
        return self.visual(image)

    def encode_text(self, text):
        B, T = text.shape
        assert T == self.context_length, 'text length %d != context_length %d' % (T, self.context_length)
        text = text.contiguous().long()
        x = _EmbedFn.apply(text, self.token_embedding.weight, self.positional_embedding, self.dtype)
        x = self.transformer(x, B, T)
        # ln_final is row-wise: normalising only the EOT rows equals normalising all and selecting
        x = self.ln_final(nn.gather_rows(x, ops.eot_index(text)))
        return self.text_projection(x, out_f32=True)

    @staticmethod
    def _tower_overlap(image):
        return (os.environ.get('PASSL_CLIP_TOWER_OVERLAP', '1') != '0' and torch.is_grad_enabled() and
                streams.enabled(image) and not torch.cuda.is_current_stream_capturing())

    def clip_logit_scale(self):
        """clip.py:309-311 — performed inside the logits kernel sequence, right after exp(s) is taken."""

    def forward(self, image, text, is_train=True, multi_rank=False):
        """multi_rank (an extension; the reference computes the loss over the local batch only): logits
        against the gathered features of every rank -> (image_logits, text_logits) of shape [B, W*B]."""
        if not is_train:
            raise NotImplementedError('is_train=False (unit logit scale) is an evaluation path')
        if self._tower_overlap(image):
            # The two towers share nothing until the logits: the text tower (77 tokens: GEMMs that fill a fraction of the
            # CUs) runs on a stream of its own next to the image tower, forward here and — autograd runs a node on the
            # stream of its forward — backward as well.  Same kernels, same order within each tower: same bits.
            dev = image.device
            main, side = torch.cuda.current_stream(dev), streams.key_stream(dev)
            streams.wait_event(side, streams.record_event(main))      # tokens, refreshed weights: ready on main
            with torch.cuda.stream(side):
                text_features = _OffMainBoundary.apply(self.encode_text(text))
            image_features = self.encode_image(image)
            streams.wait_stream(main, side)
        else:
            image_features = self.encode_image(image)
            text_features = self.encode_text(text)
        if multi_rank:
            return _CrossRankLogitsFn.apply(image_features, text_features, self.logit_scale)
        image_logits = _LogitsFn.apply(image_features, text_features, self.logit_scale)
        return image_logits, image_logits.t()
