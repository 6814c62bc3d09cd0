"""OptimizerHook: clear_grad -> backward -> step, reference passl_v110/hooks/optimizer_hook.py:25-50
(the non-AMP, non-LARS branch is the MoCo path; bf16 needs no loss scaling)."""
from .builder import HOOKS
from .hook import Hook


@HOOKS.register()
class OptimizerHook(Hook):
    def __init__(self, priority=1):
        self.priority = priority

    def train_iter_end(self, trainer):
        trainer.optimizer.clear_grad()
        loss = trainer.outputs['loss']
        reducer = getattr(trainer, 'grad_reducer', None)
        if reducer is not None:
            reducer.begin()
        loss.backward()
        if 'lars' in trainer.optimizer.type:
            raise NotImplementedError('LARS (SimCLR) is a later scope row')
        trainer.optimizer.step()
        if 'loss' not in trainer.outputs:
            trainer.outputs['loss'] = loss
