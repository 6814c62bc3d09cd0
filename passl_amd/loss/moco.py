"""``passl.loss.moco`` — the MoCo InfoNCE loss (BASELINE.json north_star: "passl/loss/{nt_xent,moco,mae}";
the reference computes it in passl_v110/modeling/architectures/moco.py:178-180 +
passl_v110/modeling/heads/contrastive_head.py:37-78).

    loss, acc1, acc5 = info_nce(q, k, queue, T)

q, k: [N,128] L2-normalised fp32 rows; queue: [128,K] fp32 (dim-major, as the reference stores it).
ONE fused HIP forward (positive + K negatives, online log-sum-exp, rank of the positive; the
[N, K+1] logits never reach HBM) and one fused backward (csrc/head.hip), both free of atomics."""
import torch
from torch.autograd import Function

from ..hip import ops


class _InfoNCEFn(Function):
    """loss = mean_i CE([q_i.k_i | q_i.queue] / T, label 0); returns (loss[1], acc1[1], acc5[1])."""

    @staticmethod
    def forward(ctx, q, k, queue, T):
        # (undefined gradients of the non-differentiable outputs stay None: materialising them would be two ATen fill
        # kernels per step inside a step that a native plan replays, hip/replay.py)
        ctx.set_materialize_grads(False)
        out, lse, _ = ops.infonce_fwd(q.contiguous(), k.contiguous(), queue, T, want_logits=False)
        ctx.save_for_backward(q, k, queue, lse)
        ctx.T = T
        loss, acc1, acc5 = out[0:1], out[1:2], out[2:3]
        ctx.mark_non_differentiable(acc1, acc5)
        return loss, acc1, acc5

    @staticmethod
    def backward(ctx, gloss, _g1, _g5):
        q, k, queue, lse = ctx.saved_tensors
        if gloss is None:
            return None, None, None, None
        dq = ops.infonce_bwd(q, k, queue, lse, gloss.contiguous().float(), ctx.T)
        return dq, None, None, None


def info_nce(q, k, queue, T):
    return _InfoNCEFn.apply(q, k, queue, float(T))


class MoCoLoss(torch.nn.Module):
    """Module spelling: ``MoCoLoss(T)(q, k, queue) -> dict(loss, acc1, acc5)``."""

    def __init__(self, temperature=0.2):
        super().__init__()
        self.temperature = temperature

    def forward(self, q, k, queue):
        loss, acc1, acc5 = info_nce(q, k, queue, self.temperature)
        return dict(loss=loss, acc1=acc1, acc5=acc5)
