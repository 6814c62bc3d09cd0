"""v2 façade: ``ContrastiveLearningTrainingEpochLoop.train_one_step(batch) -> (None, loss_dict)``.

The symbol BASELINE.json's north_star calls "passl/engine.train_one_step" is
passl/engine/loops/contrastive_learning_loop.py:67-88 in the reference: drop the label, split the
batch into ``accum_steps`` micro-batches, forward + backward each, ``grad_sync`` (blocking
all-reduce, /nranks), optimizer step, clear_grad, lr step.  This class keeps that name, argument
and return convention on top of the same HIP model/optimizer objects the v110 Trainer uses.
"""
import collections

import torch

from ...core.sync_utils import grad_sync
from ...models import Model
from .loop import TrainingEpochLoop


class ContrastiveLearningTrainingEpochLoop(TrainingEpochLoop):
    def __init__(self, trainer, epochs=1, max_train_step=None, val_loop=None):
        """``trainer`` needs: model, optimizer, accum_steps (default 1), lr_decay_unit (the v2 Engine,
        engine/engine.py, or any object with those attributes)."""
        super().__init__(trainer, epochs, max_train_step=max_train_step, val_loop=val_loop)

    def forward_backward(self, batch):
        accum = getattr(self.trainer, 'accum_steps', 1)
        self.batch_size = batch[0].shape[0]
        assert self.batch_size % accum == 0, \
            'Bad accum_steps {} for batch size {}'.format(accum, self.batch_size)
        step = self.batch_size // accum
        final = collections.defaultdict(float)
        for idx in range(accum):
            sub = [b[idx * step:(idx + 1) * step] for b in batch]
            # v2 models take the sub-batch as ONE list argument (contrastive_learning_loop.py:52);
            # v110 architectures (mode='train') take the views positionally
            loss_dict = self.trainer.model(sub) if isinstance(self.trainer.model, Model) \
                else self.trainer.model(*sub)
            if torch.is_tensor(loss_dict):
                loss_dict = {'loss': loss_dict}
            loss_dict = {k: v for k, v in loss_dict.items() if torch.is_tensor(v)}
            for k in loss_dict:
                loss_dict[k] = loss_dict[k] / accum
                with torch.no_grad():
                    final[k] = final[k] + loss_dict[k].detach()
            reducer = getattr(self.trainer, 'grad_reducer', None)
            if reducer is not None and idx == accum - 1:
                reducer.begin()        # overlap the all-reduce with the LAST micro-batch's backward
            loss_dict['loss'].backward()
        return final

    def train_one_step(self, batch):
        batch = batch[0]                        # remove label  (loop.py:69)
        loss_dict = self.forward_backward(batch)
        opt = self.trainer.optimizer
        if getattr(self.trainer, 'grad_reducer', None) is None:
            grad_sync([{'params': opt._parameter_list}])      # blocking, as the reference (sync_utils.py:18-43)
        opt.step()                                            # (waits for an overlapped reducer itself)
        opt.clear_grad()
        if getattr(self.trainer, 'lr_decay_unit', 'step') == 'step':
            sched = getattr(self.trainer, 'lr_scheduler', None)
            if sched is not None:
                sched.step(self.global_step)          # optimizer.lr_step(self.global_step), loop.py:86-87
        return None, loss_dict
