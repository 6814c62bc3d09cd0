"""ClasHead — reference passl_v110/modeling/heads/clas_head.py:22-72: AdaptiveAvgPool2D(1) -> reshape
-> Linear(in_channels, num_classes) (Normal(0, 0.01), zero bias); ``loss`` = CrossEntropyLoss +
``accuracy(topk=(1, 5))`` in percent.

HIP execution: NHWC average pool, fc as the implicit-GEMM kernel with fp32 scores, cross-entropy +
top-1/top-5 by rank counting in one kernel (csrc/clas.hip)."""
import torch
from torch.autograd import Function

from ...hip import nn, ops
from .builder import HEADS


class _SoftmaxCEFn(Function):
    @staticmethod
    def forward(ctx, scores, labels):
        ctx.set_materialize_grads(False)        # (see loss/moco.py)
        scores = scores.contiguous()
        out, lse = ops.softmax_ce_fwd(scores, labels)
        ctx.save_for_backward(scores, lse, labels)
        loss, acc1, acc5 = out[0:1], out[1:2], out[2:3]
        ctx.mark_non_differentiable(acc1, acc5)
        return loss, acc1, acc5

    @staticmethod
    def backward(ctx, gloss, _g1, _g5):
        scores, lse, labels = ctx.saved_tensors
        if gloss is None:
            return None, None
        return ops.softmax_ce_bwd(scores, lse, labels, gloss.contiguous().float()), None


def accuracy(output, target, topk=(1, 5)):
    """clas_head.py:58-72 for topk = (1, 5): percentages as 1-element tensors."""
    if tuple(topk) != (1, 5):
        raise NotImplementedError('accuracy is built for topk=(1, 5)')
    with torch.no_grad():
        _loss, acc1, acc5 = _SoftmaxCEFn.apply(output.detach().float(), target.contiguous().long().view(-1))
    return [acc1, acc5]


@HEADS.register()
class ClasHead(nn.Layer):
    """Simple classifier head."""

    def __init__(self, with_avg_pool=False, in_channels=2048, num_classes=1000):
        super(ClasHead, self).__init__()
        if num_classes % 8:
            raise NotImplementedError('the GEMM kernels write 16-byte rows: num_classes must be a multiple of 8 '
                                      '(got %d)' % num_classes)
        self.with_avg_pool = with_avg_pool
        self.in_channels = in_channels
        self.num_classes = num_classes
        if self.with_avg_pool:
            self.avg_pool = nn.AdaptiveAvgPool2D((1, 1))
        self.fc_cls = nn.Linear(in_channels, num_classes)
        with torch.no_grad():                              # normal_init(fc_cls, mean 0, std 0.01, bias 0)
            self.fc_cls.weight.copy_(torch.randn(in_channels, num_classes) * 0.01)
            self.fc_cls.bias.zero_()

    def forward(self, x):
        if self.with_avg_pool:
            x = self.avg_pool(x)                           # [N,H,W,C] -> [N,C]
        x = x.reshape(-1, self.in_channels)
        return self.fc_cls(x, out_f32=True)

    def loss(self, cls_score, labels):
        losses = dict()
        loss, acc1, acc5 = _SoftmaxCEFn.apply(cls_score, labels.contiguous().long().view(-1))
        losses['loss'] = loss
        losses['acc1'], losses['acc5'] = acc1, acc5
        return losses


@HEADS.register()
class VisionTransformerClsHead(ClasHead):
    """Vision Transformer classifier head — reference passl_v110/modeling/heads/vision_transformer_head.py:23-60:
    ``fc_cls`` = Linear(in_channels, num_classes), Normal(0, 0.01) / zero bias; loss = cross-entropy + top-1 / top-5.
    (``hidden_dim`` other than None leaves the reference's head without its ``fc_cls``: refused here.)"""

    def __init__(self, with_avg_pool=False, in_channels=2048, num_classes=1000, hidden_dim=None):
        if hidden_dim is not None:
            raise NotImplementedError('VisionTransformerClsHead(hidden_dim=...) defines no classifier in the reference')
        super().__init__(with_avg_pool=with_avg_pool, in_channels=in_channels, num_classes=num_classes)
