"""OptimizerHook: clear_grad -> backward -> step, reference passl_v110/hooks/optimizer_hook.py:25-50
(non-AMP branches: ``step()`` for Momentum — MoCo — and ``minimize(loss)`` for LARS — SimCLR;
bf16 needs no loss scaling)."""
from .builder import HOOKS
from .hook import Hook


@HOOKS.register()
class OptimizerHook(Hook):
    def __init__(self, priority=1):
        self.priority = priority

    def train_iter_end(self, trainer):
        if getattr(trainer, '_step_done', False):
            # Trainer.train_step already ran forward + this hook's body as one unit (hip/graph.py: the whole
            # step is captured in a HIP graph and replayed)
            trainer._step_done = False
            return
        self.optimize(trainer)

    def optimize(self, trainer):
        if 'Lars' in trainer.cfg['optimizer']['name']:
            trainer.optimizer.clear_gradients()
        else:
            trainer.optimizer.clear_grad()
        loss = trainer.outputs['loss']
        reducer = getattr(trainer, 'grad_reducer', None)
        if reducer is not None:
            reducer.begin()
        # (an explicit root gradient: autograd's own ones_like would be a fill kernel per step that a step plan —
        # hip/replay.py — cannot replay)
        if loss.is_cuda:
            from ..hip import ops
            loss.backward(ops.ones_like_cached(loss))
        else:
            loss.backward()
        if 'lars' in trainer.optimizer.type:
            trainer.optimizer.minimize(loss)
        else:
            trainer.optimizer.step()
        if 'loss' not in trainer.outputs:
            trainer.outputs['loss'] = loss
