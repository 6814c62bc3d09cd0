"""Momentum-SGD over the flat parameter arena (one HIP launch per step).

Registered under the reference's name ``Momentum`` (passl_v110/solver/optimizer.py:24 registers
paddle.optimizer.Momentum) with Paddle's constructor spelling
``Momentum(learning_rate, momentum=0.9, parameters=None, weight_decay=None, ...)``.  A float
``weight_decay`` is L2 decay folded into the gradient of EVERY parameter (configs/moco has no
exclusion list): g += wd*p; v = mu*v + g; p -= lr*v  — the rule restated in-tree at
passl/optimizer/momentum.py:150-158.
"""
import torch

from ..hip import ops
from .builder import OPTIMIZERS
from .lr_scheduler import LRScheduler


@OPTIMIZERS.register()
class Momentum(object):
    type = 'momentum'

    def __init__(self, learning_rate=0.001, momentum=0.9, parameters=None, use_nesterov=False,
                 weight_decay=None, grad_clip=None, multi_precision=False, rescale_grad=1.0,
                 name=None):
        if use_nesterov:
            raise NotImplementedError('nesterov momentum is not used on the MoCo path')
        if grad_clip is not None:
            raise NotImplementedError('grad_clip is not used on the MoCo path')
        self._learning_rate = learning_rate
        self._momentum = float(momentum)
        self._wd = float(weight_decay) if weight_decay else 0.0
        self._rescale = float(rescale_grad)
        self._parameter_list = [p for p in (parameters or []) if p.requires_grad]
        arenas = []
        for p in self._parameter_list:
            a = getattr(p, '_passl_arena', None)
            if a is None:
                raise NotImplementedError('Momentum optimises parameters that live in an '
                                          'EncoderArena (flat buffer); got a free tensor')
            if a not in arenas:
                arenas.append(a)
        for a in arenas:
            n_listed = sum(1 for p in self._parameter_list if p._passl_arena is a)
            if n_listed != len(a.param_slices):
                raise NotImplementedError('optimising a subset of an arena is not supported')
        self._arenas = arenas
        self._velocity = [torch.zeros_like(a.flat[:a.n_train]) for a in arenas]
        self.grad_scale = 1.0      # set by the DP reducer to 1/world_size (sum -> mean)

    # ---- paddle.optimizer API used by the hooks
    def get_lr(self):
        lr = self._learning_rate
        return float(lr()) if isinstance(lr, LRScheduler) else float(lr)

    def clear_grad(self, set_to_zero=True):
        for a in self._arenas:
            a.clear_grad()

    clear_gradients = clear_grad

    @torch.no_grad()
    def step(self):
        lr = self.get_lr()
        for a, v in zip(self._arenas, self._velocity):
            if a.reducer is not None:
                a.reducer.finish()
            ops.momentum_sgd(a.flat[:a.n_train], a.grads, v, lr, self._momentum, self._wd,
                             self.grad_scale * self._rescale)

    def state_dict(self):
        sd = {'velocity_%d' % i: v.detach().cpu() for i, v in enumerate(self._velocity)}
        if isinstance(self._learning_rate, LRScheduler):
            sd['LR_Scheduler'] = self._learning_rate.state_dict()
        return sd

    def set_state_dict(self, sd):
        for i, v in enumerate(self._velocity):
            v.copy_(sd['velocity_%d' % i])
        if 'LR_Scheduler' in sd and isinstance(self._learning_rate, LRScheduler):
            self._learning_rate.set_state_dict(sd['LR_Scheduler'])
