"""SimCLRContrastiveHead — NT-Xent over [ab|aa] / [ba|bb] plus 3 x CO2 consistency (two KL terms),
reference passl_v110/modeling/heads/simclr_contrastive_head.py:25-102; ONE fused HIP kernel each
way (passl_amd/csrc/ntxent.hip): the B x 2B logit matrices never reach HBM.

``multi_rank`` is stored and ignored by the reference (hidden*_large = hidden*, lines 55-56), so
the default reproduces it exactly.  ``multi_rank=True`` here enables the cross-GPU negative set of
BASELINE configs[2] (an extension, defined as in passl/models/mocov3.py:187-198: all-gather the
embeddings, positives offset by rank*N; the gradient of the gathered copies is reduce-scattered
back)."""
from ...hip import nn
from .builder import HEADS


from ...loss.nt_xent import _NTXentFn     # the fused loss lives in passl.loss.nt_xent


@HEADS.register()
class SimCLRContrastiveHead(nn.Layer):
    def __init__(self, temperature=0.5, return_accuracy=True, multi_rank=False, co2_weight=3.0):
        super().__init__()
        self.temperature = temperature
        self.return_accuracy = return_accuracy
        self.multi_rank = multi_rank
        self.co2_weight = co2_weight

    def forward(self, pos, neg):
        """pos, neg: the two views' embeddings hidden1, hidden2 [N, 128] fp32 (the reference keeps
        the argument names of ContrastiveHead)."""
        loss, acc1 = _NTXentFn.apply(pos, neg, float(self.temperature), float(self.co2_weight),
                                     bool(self.multi_rank))
        outputs = dict(loss=loss)
        outputs['acc1'] = acc1
        return outputs
