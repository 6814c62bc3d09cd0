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


class ContrastiveLearningTrainingEpochLoop(object):
    def __init__(self, trainer, epochs=1, max_train_step=None, val_loop=None):
        """``trainer`` needs: model, optimizer, accum_steps (default 1), lr_decay_unit."""
        self.trainer = trainer
        self.epochs = epochs
        self.max_train_step = max_train_step
        self.global_step = 0

    def forward_backward(self, batch):
        accum = getattr(self.trainer, 'accum_steps', 1)
        self.batch_size = batch[0].shape[0]
        assert self.batch_size % accum == 0, \
            'Bad accum_steps {} for batch size {}'.format(accum, self.batch_size)
        step = self.batch_size // accum
        final = collections.defaultdict(float)
        for idx in range(accum):
            sub = [b[idx * step:(idx + 1) * step] for b in batch]
            loss_dict = self.trainer.model(*sub)
            if torch.is_tensor(loss_dict):
                loss_dict = {'loss': loss_dict}
            for k in loss_dict:
                loss_dict[k] = loss_dict[k] / accum
                with torch.no_grad():
                    final[k] = final[k] + loss_dict[k].detach()
            loss_dict['loss'].backward()
        return final

    def train_one_step(self, batch):
        batch = batch[0]                        # remove label  (loop.py:69)
        loss_dict = self.forward_backward(batch)
        opt = self.trainer.optimizer
        grad_sync([{'params': opt._parameter_list}])
        opt.step()
        opt.clear_grad()
        if getattr(self.trainer, 'lr_decay_unit', 'step') == 'step':
            sched = getattr(self.trainer, 'lr_scheduler', None)
            if sched is not None:
                sched.step()
        self.global_step += 1
        return None, loss_dict
