"""CELoss — reference passl/loss/celoss.py:22-56: mean softmax cross entropy over hard labels, returned as
``{"CELoss": loss}``.  One kernel (csrc/clas.hip: row log-sum-exp, the label's score and the top-1 / top-5 ranks in
one pass; backward = softmax - one_hot scaled by the incoming gradient).  Label smoothing (``epsilon``) and soft
labels — the fine-tuning recipes' mixup targets — are not on the linear-probe path and raise."""
import torch

from ..hip import nn as hnn
from ..modeling.heads.clas_head import _SoftmaxCEFn


class CELoss(hnn.Layer):
    """Softmax Cross entropy loss"""

    def __init__(self, epsilon=None):
        super().__init__()
        if epsilon is not None:
            assert epsilon >= 0 and epsilon <= 1, 'epsilon must be in [0, 1]'
            raise NotImplementedError('label smoothing (CELoss epsilon) is used by the fine-tuning recipes only')
        self.epsilon = epsilon

    def forward(self, x, label):
        if isinstance(x, dict):
            x = x['logits']
        if label.dim() > 1 and label.shape[-1] == x.shape[-1]:
            raise NotImplementedError('soft labels (mixup / cutmix targets) are used by the fine-tuning recipes only')
        loss, _acc1, _acc5 = _SoftmaxCEFn.apply(x.float(), label.contiguous().long().view(-1))
        return {'CELoss': loss.reshape(())}
