"""Cosine-momentum average of a model — reference passl/models/utils/averaged_model.py:26-188
(BaseAveragedModel / ExponentialMovingAverage / CosineEMA), as used by MoCo-v3 and BYOL-style methods.

    update_parameters:   steps == 0   avg <- source                               (:72-74)
                         otherwise    avg <- avg*(1 - m_t) + source*m_t           (:186-188)
                                      m_t = end - (end - momentum)*(cos(pi*steps/max_steps) + 1)/2
                         steps += 1

``max_steps`` defaults to ``runtime_info_hub.max_steps`` (filled by the Engine).  The averaged copy and the source
live in two EncoderArenas of identical layout, so one update is ONE kernel launch over the flat buffers (weights and
BatchNorm running statistics: Paddle keeps the statistics as non-trainable parameters, i.e. inside
``named_parameters()``, which is what the reference averages — there are no separate buffers to copy).  ``steps`` is
a device buffer with the reference's state_dict key; a host mirror avoids a device read per iteration.
"""
import math

import torch

from ...hip import config
from ...hip import nn as hnn
from ...utils.infohub import runtime_info_hub


class CosineEMA(hnn.Layer):
    def __init__(self, model, max_steps=None, momentum=0.004, end_momentum=0., interval=1, update_buffers=False):
        super().__init__()
        assert 0.0 < momentum < 1.0, 'momentum must be in range (0.0, 1.0) but got {}'.format(momentum)
        if update_buffers:
            raise NotImplementedError('update_buffers=True (no layer of the MoCo-v3 model has buffers in Paddle)')
        self.model = model
        for p in self.model.parameters():
            p.requires_grad_(False)
        self.interval = interval
        self.momentum, self.end_momentum, self.max_steps = momentum, end_momentum, max_steps
        self.register_buffer('steps', torch.zeros((), dtype=torch.int64, device=config.get_device()))
        self._steps = 0
        self._avg = self._src = None

    def bind(self, avg_arena, src_arena):
        self._avg, self._src = avg_arena, src_arena

    def sync_steps(self):
        """After the `steps` buffer was loaded / broadcast."""
        self._steps = int(self.steps)

    def forward(self, *args, **kwargs):
        return self.model(*args, **kwargs)

    def current_momentum(self, steps=None):
        steps = self._steps if steps is None else steps
        max_steps = self.max_steps if self.max_steps is not None else runtime_info_hub.max_steps
        cosine_annealing = (math.cos(math.pi * steps / float(max_steps)) + 1) / 2
        return self.end_momentum - (self.end_momentum - self.momentum) * cosine_annealing

    @torch.no_grad()
    def update_parameters(self, model=None):
        """``model`` is accepted for API parity; the source is the arena given to ``bind``."""
        if self._avg is None:
            raise RuntimeError('CosineEMA.bind(avg_arena, src_arena) was not called')
        if self._steps == 0:
            self._avg.copy_from(self._src)
        elif self._steps % self.interval == 0:
            m = self.current_momentum()
            self._avg.ema_from(self._src, 1.0 - m)          # avg*(1-m) + src*m
        self._steps += 1
        self.steps.add_(1)
