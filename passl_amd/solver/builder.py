"""build_lr_scheduler / build_optimizer — reference passl_v110/solver/builder.py:26-216."""
import copy

from ..utils.registry import Registry, build_from_config

LRSCHEDULERS = Registry('LRSCHEDULER')
OPTIMIZERS = Registry('OPTIMIZER')


def build_lr_scheduler(cfg, iters_per_epoch):
    if cfg.name in ('CosineAnnealingDecay',):
        cfg.T_max *= iters_per_epoch          # yaml T_max is in epochs (builder.py:28-30)
        return build_from_config(cfg, LRSCHEDULERS)
    elif cfg.name == 'MultiStepDecay':
        cfg.milestones = [x * iters_per_epoch for x in cfg.milestones]
        return build_from_config(cfg, LRSCHEDULERS)
    elif cfg.name == 'LinearWarmup':
        cfg.learning_rate = build_lr_scheduler(cfg.learning_rate, iters_per_epoch)
        cfg.warmup_steps *= iters_per_epoch
        return build_from_config(cfg, LRSCHEDULERS)
    raise NotImplementedError(cfg.name)


def build_optimizer(cfg, lr_scheduler, model_list=None):
    cfg = copy.deepcopy(cfg)
    name = cfg.pop('name')
    if 'layer_decay' in cfg and float(cfg.pop('layer_decay')) < 1.0:
        raise NotImplementedError('layer-wise lr decay (ViT fine-tuning) is outside the MoCo hot path')
    if 'grad_clip' in cfg:
        raise NotImplementedError('grad_clip is not used by configs/moco and is not built')
    parameters = sum([list(m.parameters()) for m in model_list], []) if model_list else None
    if 'Lars' in name or 'Lamb' in name:
        cfg['parameter_list'] = parameters
    else:
        cfg['parameters'] = parameters
    return OPTIMIZERS.get(name)(lr_scheduler, **cfg)
