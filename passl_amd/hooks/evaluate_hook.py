"""EvaluateHook — reference passl_v110/hooks/evaluate_hook.py:25-41: ``trainer.val(**eval_kargs)`` at the
end of every training epoch (and before the first one with ``init_eval``)."""
from .hook import Hook
from .builder import HOOKS


@HOOKS.register()
class EvaluateHook(Hook):
    def __init__(self, init_eval=False, eval_kargs=None, priority=1):
        self.eval_kargs = {} if eval_kargs is None else eval_kargs
        self.init_eval = init_eval
        self.priority = priority

    def run_begin(self, trainer):
        if self.init_eval:
            trainer.val(**self.eval_kargs)

    def train_epoch_end(self, trainer):
        trainer.val(**self.eval_kargs)
