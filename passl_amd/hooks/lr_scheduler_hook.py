"""LRSchedulerHook — steps ``trainer.lr_scheduler`` once per iteration (default) or once per epoch
(reference passl_v110/hooks/lr_scheduler_hook.py:19-32; the MoCo / SimCLR / MAE / CLIP configs use the
per-iteration unit, their schedulers are built in iteration units by ``build_lr_scheduler``)."""
from .builder import HOOKS
from .hook import Hook

_UNITS = ('iter', 'epoch')


@HOOKS.register()
class LRSchedulerHook(Hook):
    def __init__(self, unit='iter', priority=1):
        assert unit in _UNITS, 'unit must be one of %s' % (_UNITS,)
        self.unit = unit
        self.priority = priority

    def _step_if(self, trainer, unit):
        if unit == self.unit:
            trainer.lr_scheduler.step()

    def train_iter_end(self, trainer):
        self._step_if(trainer, 'iter')

    def train_epoch_end(self, trainer):
        self._step_if(trainer, 'epoch')
