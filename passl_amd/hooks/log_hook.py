"""LogHook — same log lines and AverageMeter bookkeeping as the reference
(passl_v110/hooks/log_hook.py:25-160) with ONE difference: the reference converts every output
with ``float(v)`` on every iteration, i.e. a device->host sync per step.  Here the
1-element device tensors are kept and flushed to the meters in a single transfer when a line is
printed (or at epoch end), so the step loop never blocks on the GPU between prints."""
import datetime
import os.path as osp
from collections import OrderedDict

import torch

from ..utils import AverageMeter
from .builder import HOOKS
from .hook import Hook


@HOOKS.register()
class LogHook(Hook):
    def __init__(self, by_epoch=True, interval=10, ignore_last=True, reset_flag=True, priority=1):
        self.interval = interval
        self.ignore_last = ignore_last
        self.reset_flag = reset_flag
        self.by_epoch = by_epoch
        self.time_sec_tot = 0
        self.priority = priority
        self._pending = OrderedDict()

    def run_begin(self, trainer):
        self.start_iter = trainer.current_iter
        self.json_log_path = osp.join(trainer.output_dir, '{}.log.json'.format(trainer.timestamp))

    def _flush(self, trainer):
        if not self._pending:
            return
        keys = list(self._pending)
        dev = [torch.stack([t.detach().reshape(()).float() for t in self._pending[k]]) for k in keys]
        host = torch.cat(dev).cpu().tolist() if dev else []
        i = 0
        for k in keys:
            n = len(self._pending[k])
            if k not in trainer.logs:
                trainer.logs[k] = AverageMeter(k, ':.4e' if 'loss' in k else ':6.3f')
            for v in host[i:i + n]:
                trainer.logs[k].update(v)
            i += n
        self._pending.clear()

    _META_KEYS = frozenset(('mode', 'Epoch', 'epoch', 'iter', 'lr', 'time', 'data_time', 'memory'))

    def _prefix(self, logs, trainer):
        """The part of the line before the meters (same text as the reference's log lines)."""
        if trainer.mode != 'train':
            return 'Epoch({}) [{}][{}]\t'.format(logs['mode'], logs['epoch'] - 1, logs['iter'])
        where = ('Epoch [{}/{}][{}/{}]\t'.format(logs['epoch'], trainer.epochs, logs['iter'], trainer.iters_per_epoch)
                 if self.by_epoch else 'Iter [{}/{}]\t'.format(logs['iter'], trainer.total_iters))
        parts = [where + 'lr: {:.3e}'.format(logs['lr'])]
        timer = logs.get('time')
        if timer is not None:
            self.time_sec_tot += timer.sum
            remaining = trainer.total_iters - trainer.current_iter - 1
            parts.append('eta: {}'.format(datetime.timedelta(seconds=int(timer.avg * remaining))))
            parts.append('time: {:.3f}, data_time: {:.3f}'.format(timer.avg, logs['data_time'].avg))
        return ', '.join(parts) + ', '

    def print_log(self, trainer):
        self._flush(trainer)
        logs = trainer.logs
        lr = trainer.lr_scheduler.get_lr()
        logs.update(mode='train' if 'time' in logs else 'val', epoch=trainer.current_epoch + 1,
                    iter=trainer.inner_iter if self.by_epoch else trainer.current_iter,
                    lr=lr[0] if isinstance(lr, list) else lr)
        meters = [str(v) for k, v in logs.items() if k not in self._META_KEYS]
        trainer.logger.info(self._prefix(logs, trainer) + ', '.join(meters))

    def epoch_begin(self, trainer):
        self._pending.clear()
        trainer.logs.clear()

    def train_iter_end(self, trainer):
        for k, v in trainer.outputs.items():
            if torch.is_tensor(v):
                # detach: a queued loss must not keep its autograd graph alive until the next print
                self._pending.setdefault(k, []).append(v.detach())
            else:
                if k not in trainer.logs:
                    trainer.logs[k] = AverageMeter(k, ':.4e' if 'loss' in k else ':6.3f')
                trainer.logs[k].update(float(v))
        if self.by_epoch and self.every_n_inner_iters(trainer, self.interval):
            self.print_log(trainer)
        elif self.every_n_iters(trainer, self.interval):
            # iteration-based runs print nothing here (as the reference), but the queue is still
            # drained every `interval` iterations so it cannot grow for a whole epoch
            self._flush(trainer)

    def train_epoch_end(self, trainer):
        self._flush(trainer)
        if self.reset_flag:
            trainer.logs.clear()
