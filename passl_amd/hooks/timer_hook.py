"""IterTimerHook — wall-clock bookkeeping for the log line (reference passl_v110/hooks/timer_hook.py:24-39):
``logs['data_time']`` = time from the end of the previous iteration (or the epoch start) to the start of
this one, ``logs['time']`` = full iteration time; both are AverageMeters read by LogHook."""
import time

from ..utils import AverageMeter
from .builder import HOOKS
from .hook import Hook


@HOOKS.register()
class IterTimerHook(Hook):
    def __init__(self, priority=1):
        self.priority = priority
        self._mark = None

    def _elapsed(self):
        return time.time() - self._mark

    @staticmethod
    def _meter(trainer, key):
        meter = trainer.logs.get(key)
        if meter is None:
            meter = trainer.logs[key] = AverageMeter(key)
        return meter

    def epoch_begin(self, trainer):
        self._mark = time.time()

    def iter_begin(self, trainer):
        self._meter(trainer, 'data_time').update(self._elapsed())

    def iter_end(self, trainer):
        self._meter(trainer, 'time').update(self._elapsed())
        self._mark = time.time()
