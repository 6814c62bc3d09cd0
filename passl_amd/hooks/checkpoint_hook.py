"""CheckpointHook — rank-0 ``epoch_N.pd`` pickles of {epoch, state_dict, optimizer, lr_scheduler}
as numpy arrays + ``latest.pd`` link, keeping at most ``max_keep_ckpts`` files; same layout idea
as the reference (passl_v110/hooks/checkpoint_hook.py:23-141).  Scope row §8f-3 ("next")."""
import os
import pickle

from .builder import HOOKS
from .hook import Hook


from ..utils.checkpoint import to_numpy as _to_numpy


def save_checkpoint(path, trainer):
    sd = {'epoch': trainer.current_epoch + 1,
          'state_dict': _to_numpy(dict(trainer.model.state_dict())),
          'optimizer': _to_numpy(trainer.optimizer.state_dict()),
          'lr_scheduler': trainer.lr_scheduler.state_dict()}
    with open(path, 'wb') as f:
        pickle.dump(sd, f, protocol=2)


@HOOKS.register()
class CheckpointHook(Hook):
    def __init__(self, interval=1, by_epoch=True, save_optimizer=True, out_dir=None,
                 max_keep_ckpts=5, priority=1, **kwargs):
        self.interval = interval
        self.by_epoch = by_epoch
        self.out_dir = out_dir
        self.max_keep_ckpts = max_keep_ckpts
        self.priority = priority

    def train_epoch_end(self, trainer):
        if not self.by_epoch or self.interval <= 0 or not self.every_n_epochs(trainer, self.interval):
            return
        if getattr(trainer, 'rank', 0) != 0:
            return
        out_dir = self.out_dir or trainer.output_dir
        os.makedirs(out_dir, exist_ok=True)
        path = os.path.join(out_dir, 'epoch_{}.pd'.format(trainer.current_epoch + 1))
        save_checkpoint(path, trainer)
        latest = os.path.join(out_dir, 'latest.pd')
        if os.path.lexists(latest):
            os.remove(latest)
        os.symlink(os.path.basename(path), latest)
        if self.max_keep_ckpts > 0:
            old = trainer.current_epoch + 1 - self.max_keep_ckpts * self.interval
            p = os.path.join(out_dir, 'epoch_{}.pd'.format(old))
            if old > 0 and os.path.exists(p):
                os.remove(p)
