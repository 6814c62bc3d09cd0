"""ContrastiveHead (InfoNCE + top-1/top-5) — reference passl_v110/modeling/heads/
contrastive_head.py:21-78.

``fused(q, k, queue)`` is the hot-path entry used by MoCo.train_iter: one HIP kernel computes the
positive logit, the 65 536 negative logits, the online log-sum-exp and the rank of the positive
without materialising the [N, K+1] logits (passl_amd/csrc/head.hip).  ``forward(pos, neg)`` keeps
the reference's call signature for callers that already hold logits; it is not on the hot path: the
materialised [N, 1 + K] logits go through the row cross-entropy / rank-counting kernel of the linear probe
(csrc/clas.hip) — the only torch ops are the concatenation and the 1/T scale of its input.
"""
import torch

from ...hip import nn
from .builder import HEADS
from .clas_head import _SoftmaxCEFn


@HEADS.register()
class ContrastiveHead(nn.Layer):
    def __init__(self, temperature=0.1, return_accuracy=True):
        super().__init__()
        self.temperature = temperature
        self.return_accuracy = return_accuracy

    def fused(self, q, k, queue):
        """q,k: [N,128] L2-normalised fp32; queue: [128,K] fp32 snapshot (pre-enqueue)."""
        loss, acc1, acc5 = nn.infonce(q, k, queue, self.temperature)
        outputs = dict(loss=loss)
        if self.return_accuracy:
            outputs['acc1'] = acc1
            outputs['acc5'] = acc5
        return outputs

    def forward(self, pos, neg):
        """Compatibility entry (materialised logits): pos [N,1], neg [N,K]."""
        N = pos.shape[0]
        logits = (torch.cat((pos, neg), dim=1) / self.temperature).float()
        labels = torch.zeros((N,), dtype=torch.int64, device=logits.device)
        loss, acc1, acc5 = _SoftmaxCEFn.apply(logits, labels)
        outputs = dict(loss=loss)
        if self.return_accuracy:
            outputs['acc1'], outputs['acc5'] = acc1, acc5
        return outputs
