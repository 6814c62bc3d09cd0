"""``passl.models.mae`` — the v2 masked autoencoder on the MI355X HIP path.

Names, constructor arguments, state_dict keys, the factories and the ``model(imgs, mask_ratio=0.75) -> (loss, pred,
mask)`` contract are the reference's (passl/models/mae.py: MaskedAutoencoderViT :37-290, MAEVisionTransformer
:293-329, factories :331-405, the recommended-arch aliases :408-410).  The arithmetic is the v110 twin's
(passl_v110/modeling/backbones/mae.py:318-564 — the two files state the same model), so the class is the HIP ``MAE``
backbone (passl_amd/modeling/backbones/mae.py: patch-embedding GEMM, keep-gather + class token + position add, fused
attention blocks, decoder unshuffle with mask tokens, fp32 pixel prediction, masked-patch loss) with what differs in
v2: the initialisation of the two tokens (both ``normal_(std=.02)``, :139-141; the v110 file truncates the class
token's and leaves the mask token at zero) and the ``Model`` contract (``load_pretrained`` / ``save``).  The flat
parameter arena is built by the constructor, so the object trains as it is: ``tasks/ssl/mae/engine_pretrain.py``'s loop
is mirrored in passl_amd/engine/loops/mae_pretrain_loop.py."""
import os
import pickle
from functools import partial

import torch

from ..hip import nn as hnn
from ..hip.nn import EncoderArena
from ..modeling.backbones.mae import MAE as _HipMAE
from ..utils.checkpoint import load_lenient, load_pickle, to_numpy
from .base_model import Model
from .vision_transformer import VisionTransformer

__all__ = [
    'MaskedAutoencoderViT', 'mae_vit_base_patch16', 'mae_vit_large_patch16', 'mae_vit_huge_patch14',
    'MAEVisionTransformer', 'maevit_base_patch16', 'maevit_large_patch16', 'maevit_huge_patch14'
]


class MaskedAutoencoderViT(_HipMAE, Model):
    """Masked Autoencoder with VisionTransformer backbone."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=1024, depth=24, num_heads=16,
                 decoder_embed_dim=512, decoder_depth=8, decoder_num_heads=16, mlp_ratio=4.,
                 norm_layer=partial(hnn.LayerNorm, epsilon=1e-6), norm_pix_loss=False):
        super().__init__(img_size=img_size, patch_size=patch_size, in_chans=in_chans, embed_dim=embed_dim,
                         depth=depth, num_heads=num_heads, decoder_embed_dim=decoder_embed_dim,
                         decoder_depth=decoder_depth, decoder_num_heads=decoder_num_heads, mlp_ratio=mlp_ratio,
                         norm_layer=norm_layer, norm_pix_loss=norm_pix_loss)
        with torch.no_grad():
            # mae.py:139-141 ("timm's trunc_normal_(std=.02) is effectively normal_(std=0.02) as cutoff is too big")
            self.cls_token.copy_(torch.randn(self.cls_token.shape) * 0.02)
            self.mask_token.copy_(torch.randn(self.mask_token.shape) * 0.02)
        self.arena_q = EncoderArena(self, trainable=True)         # flat parameters / gradients (AdamW, DP reducer)

    def sync_runtime_state(self):
        self.arena_q.refresh()

    def load_state_dict(self, state_dict, strict=True):
        r = super().load_state_dict(state_dict, strict=strict)
        self.sync_runtime_state()
        return r

    def forward(self, imgs, mask_ratio=0.75, noise=None):
        """-> (loss, pred [N, L, p*p*3], mask [N, L]).  ``noise``: inject the per-sample masking noise (tests)."""
        self.arena_q.refresh()                                   # compute-dtype copies of the updated weights
        return super().forward(imgs, mask_ratio, noise=noise)

    def load_pretrained(self, path, rank=0, finetune=False):
        fn = path if os.path.exists(path) else path + '.pdparams'
        if not os.path.exists(fn):
            raise ValueError('Model pretrain path {} does not exists.'.format(fn))
        load_lenient(self, load_pickle(fn), what='pretrained MAE')
        self.sync_runtime_state()

    def save(self, path, local_rank=0, rank=0):
        if rank != 0:
            return
        os.makedirs(os.path.dirname(path) or '.', exist_ok=True)
        with open(path + '.pdparams', 'wb') as f:
            pickle.dump(to_numpy(dict(self.state_dict())), f, protocol=2)


class MAEVisionTransformer(VisionTransformer):
    """Vision Transformer with support for global average pooling (the fine-tuning / linear-probe encoder,
    mae.py:293-329).  ``global_pool=True`` (mean over the patch tokens + ``fc_norm``) belongs to the fine-tuning
    recipes and is not built on the HIP path."""

    def __init__(self, global_pool=False, **kwargs):
        if global_pool:
            raise NotImplementedError('MAEVisionTransformer(global_pool=True) is a fine-tuning variant (outside the '
                                      'pre-training path)')
        super().__init__(**kwargs)
        self.global_pool = False


def mae_vit_base_patch16_dec512d8b(**kwargs):
    return MaskedAutoencoderViT(patch_size=16, embed_dim=768, depth=12, num_heads=12, decoder_embed_dim=512,
                                decoder_depth=8, decoder_num_heads=16, mlp_ratio=4,
                                norm_layer=partial(hnn.LayerNorm, epsilon=1e-6), **kwargs)


def mae_vit_large_patch16_dec512d8b(**kwargs):
    return MaskedAutoencoderViT(patch_size=16, embed_dim=1024, depth=24, num_heads=16, decoder_embed_dim=512,
                                decoder_depth=8, decoder_num_heads=16, mlp_ratio=4,
                                norm_layer=partial(hnn.LayerNorm, epsilon=1e-6), **kwargs)


def mae_vit_huge_patch14_dec512d8b(**kwargs):
    return MaskedAutoencoderViT(patch_size=14, embed_dim=1280, depth=32, num_heads=16, decoder_embed_dim=512,
                                decoder_depth=8, decoder_num_heads=16, mlp_ratio=4,
                                norm_layer=partial(hnn.LayerNorm, epsilon=1e-6), **kwargs)


def maevit_base_patch16(**kwargs):
    return MAEVisionTransformer(patch_size=16, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4, qkv_bias=True,
                                norm_layer=partial(hnn.LayerNorm, epsilon=1e-6), **kwargs)


def maevit_large_patch16(**kwargs):
    return MAEVisionTransformer(patch_size=16, embed_dim=1024, depth=24, num_heads=16, mlp_ratio=4, qkv_bias=True,
                                norm_layer=partial(hnn.LayerNorm, epsilon=1e-6), **kwargs)


def maevit_huge_patch14(**kwargs):
    return MAEVisionTransformer(patch_size=14, embed_dim=1280, depth=32, num_heads=16, mlp_ratio=4, qkv_bias=True,
                                norm_layer=partial(hnn.LayerNorm, epsilon=1e-6), **kwargs)


# set recommended archs (mae.py:408-410)
mae_vit_base_patch16 = mae_vit_base_patch16_dec512d8b  # decoder: 512 dim, 8 blocks
mae_vit_large_patch16 = mae_vit_large_patch16_dec512d8b  # decoder: 512 dim, 8 blocks
mae_vit_huge_patch14 = mae_vit_huge_patch14_dec512d8b  # decoder: 512 dim, 8 blocks
