"""TopkAcc — reference passl/metric/metrics.py:29-56: ``{"top1": a1, "metric": a1, "top5": a5}`` with
paddle.metric.accuracy's FRACTIONS in [0, 1] (the v110 heads report percentages).

Difference by design: the values stay 0-d DEVICE tensors (the reference calls ``.item()`` = one device->host sync
per metric per step); the loops convert them when a line is printed / an evaluation pass ends.  One kernel
(csrc/clas.hip) ranks the label's score in its row — ties go to the lower index, as top_k."""
import torch

from ..modeling.heads.clas_head import accuracy

__all__ = ['TopkAcc']


class TopkAcc(object):
    def __init__(self, topk=(1, 5)):
        assert isinstance(topk, (int, list, tuple))
        if isinstance(topk, int):
            topk = [topk]
        self.topk = list(topk)
        if any(k not in (1, 5) for k in self.topk):
            raise NotImplementedError('the rank kernel reports top-1 and top-5 (got topk=%r)' % (self.topk,))

    def __call__(self, x, label):
        if isinstance(x, dict):
            x = x['logits']
        acc1, acc5 = accuracy(x, label.reshape(-1), topk=(1, 5))          # percentages, 1-element tensors
        vals = {1: acc1.reshape(()) / 100.0, 5: acc5.reshape(()) / 100.0}
        metric_dict = dict()
        for i, k in enumerate(self.topk):
            metric_dict['top{}'.format(k)] = vals[k]
            if i == 0:
                metric_dict['metric'] = vals[k]
        return metric_dict
