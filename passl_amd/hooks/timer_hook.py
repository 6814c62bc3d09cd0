"""IterTimerHook — reference passl_v110/hooks/timer_hook.py:24-39."""
import time

from ..utils import AverageMeter
from .builder import HOOKS
from .hook import Hook


@HOOKS.register()
class IterTimerHook(Hook):
    def __init__(self, priority=1):
        self.priority = priority

    def epoch_begin(self, runner):
        self.t = time.time()

    def iter_begin(self, runner):
        if 'data_time' not in runner.logs:
            runner.logs['data_time'] = AverageMeter('data_time')
        runner.logs['data_time'].update(time.time() - self.t)

    def iter_end(self, runner):
        if 'time' not in runner.logs:
            runner.logs['time'] = AverageMeter('time')
        runner.logs['time'].update(time.time() - self.t)
        self.t = time.time()
