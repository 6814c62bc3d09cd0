"""``passl.loss.mae`` — MAE's masked-patch reconstruction loss, reference
passl_v110/modeling/backbones/mae.py:541-557 (= passl/models/mae.py:268-284).

    loss = masked_patch_loss(pred_rows, imgs, mask, patch_size, norm_pix_loss, denom)

pred_rows: [B*(L+1), p*p*3] fp32 prediction rows, imgs: [B,3,H,W] fp32, mask: [B,L] (1 = masked).  One wave per
patch reads the image once (optional per-patch normalisation with the unbiased variance), fused
forward and backward (csrc/vit.hip)."""
from torch.autograd import Function

from ..hip import ops


class _MAELossFn(Function):
    @staticmethod
    def forward(ctx, pred, imgs, mask, p, norm_pix, denom):
        ctx.save_for_backward(pred, imgs, mask)
        ctx.args = (p, norm_pix, denom)
        return ops.mae_loss_fwd(imgs, pred, mask, p, norm_pix, denom)

    @staticmethod
    def backward(ctx, gloss):
        pred, imgs, mask = ctx.saved_tensors
        p, norm_pix, denom = ctx.args
        return ops.mae_loss_bwd(imgs, pred, mask, gloss.contiguous().float(), p, norm_pix, denom), None, \
            None, None, None, None


def masked_patch_loss(pred_rows, imgs, mask, patch_size, norm_pix_loss=False, denom=None):
    """pred_rows: the decoder's prediction rows [B*(L+1), p*p*3] (class-token rows included, skipped by
    the kernel).  ``denom`` = number of masked patches (``mask.sum()``); pass the host-known value
    (B * (L - len_keep) for MAE's fixed-ratio masking) to avoid a device->host sync."""
    if denom is None:
        denom = float(mask.sum())
    return _MAELossFn.apply(pred_rows, imgs.contiguous().float(), mask, int(patch_size), bool(norm_pix_loss),
                            float(denom))
