"""Homes of the fused losses (``passl.loss.{moco, nt_xent, mae}``, BASELINE.json north_star / SURVEY
appendix C); the heads and backbones call into these.  ``build_loss`` / ``CombinedLoss`` = the v2 loss front end
(reference passl/loss/__init__.py:25-61): a yaml list of ``{Name: {weight: w, **kwargs}}`` entries, evaluated on
(output, target) into ``{Name: w * value, ..., "loss": sum}``."""
import copy

from . import mae, moco, nt_xent
from .moco import MoCoLoss, info_nce
from .nt_xent import nt_xent as nt_xent_loss
from .mae import masked_patch_loss
from .celoss import CELoss

_LOSSES = {'CELoss': CELoss}


class CombinedLoss(object):
    def __init__(self, config_list):
        self.loss_func = []
        self.loss_weight = []
        assert isinstance(config_list, list), 'operator config should be a list'
        for config in config_list:
            assert isinstance(config, dict) and len(config) == 1, 'yaml format error'
            name = list(config)[0]
            param = dict(config[name])
            assert 'weight' in param, 'weight must be in param, but param just contains {}'.format(param.keys())
            self.loss_weight.append(param.pop('weight'))
            if name not in _LOSSES:
                raise NotImplementedError('loss %r is not on the linear-probe path (built: %s)' % (name, sorted(_LOSSES)))
            self.loss_func.append(_LOSSES[name](**param))

    def __call__(self, input, target):
        # (the reference casts fp16 / bf16 logits to fp32 here; the classifier GEMM already writes fp32 scores)
        loss_dict = {}
        for loss_func, weight in zip(self.loss_func, self.loss_weight):
            loss = loss_func(input, target)
            loss_dict.update({key: loss[key] * weight for key in loss})
        vals = list(loss_dict.values())
        total = vals[0]
        for v in vals[1:]:
            total = total + v
        loss_dict['loss'] = total
        return loss_dict


def build_loss(config):
    return CombinedLoss(copy.deepcopy(config))
