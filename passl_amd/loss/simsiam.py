"""SimSiam's criterion — reference passl/models/simsiam.py:69,93: ``-nn.CosineSimilarity(axis=1)(p, z.detach()).mean()``
as ONE fused row kernel pair (csrc/head.hip: passl_hip_cosine_loss_fwd / _bwd); the mean is a fixed-order sum."""
from torch.autograd import Function

from ..hip import ops

__all__ = ['neg_cosine_similarity']


class _NegCosineFn(Function):
    @staticmethod
    def forward(ctx, p, z, eps):
        p, z = p.contiguous(), z.detach().contiguous()
        loss, stats = ops.cosine_loss_fwd(p, z, eps)
        ctx.save_for_backward(p, z, stats)
        return loss

    @staticmethod
    def backward(ctx, gloss):
        p, z, stats = ctx.saved_tensors
        return ops.cosine_loss_bwd(p, z, stats, gloss.contiguous().float()), None, None


def neg_cosine_similarity(p, z, eps=1e-8):
    """-mean_i cos(p_i, z_i) for fp32 rows [N, D]; z carries no gradient (stop-gradient)."""
    return _NegCosineFn.apply(p, z, eps)
