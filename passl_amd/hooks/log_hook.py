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

    def _log_info(self, log_dict, trainer):
        if trainer.mode == 'train':
            lr_str = 'lr: {:.3e}'.format(log_dict['lr'])
            if self.by_epoch:
                log_str = 'Epoch [{}/{}][{}/{}]\t'.format(log_dict['epoch'], trainer.epochs,
                                                          log_dict['iter'], trainer.iters_per_epoch)
            else:
                log_str = 'Iter [{}/{}]\t'.format(log_dict['iter'], trainer.total_iters)
            log_str += '{}, '.format(lr_str)
            if 'time' in log_dict.keys():
                self.time_sec_tot += log_dict['time'].sum
                time_sec_avg = log_dict['time'].avg
                eta_sec = time_sec_avg * (trainer.total_iters - trainer.current_iter - 1)
                log_str += 'eta: {}, '.format(str(datetime.timedelta(seconds=int(eta_sec))))
                log_str += 'time: {:.3f}, data_time: {:.3f}, '.format(time_sec_avg,
                                                                      log_dict['data_time'].avg)
        else:
            log_str = 'Epoch({}) [{}][{}]\t'.format(log_dict['mode'], log_dict['epoch'] - 1,
                                                    log_dict['iter'])
        items = []
        for name, val in log_dict.items():
            if name in ['mode', 'Epoch', 'iter', 'lr', 'time', 'data_time', 'memory', 'epoch']:
                continue
            items.append(str(val) if isinstance(val, AverageMeter) else val)
        trainer.logger.info(log_str + ', '.join(str(i) for i in items))

    def print_log(self, trainer):
        self._flush(trainer)
        log_dict = trainer.logs
        mode = 'train' if 'time' in trainer.logs else 'val'
        log_dict['mode'] = mode
        log_dict['epoch'] = trainer.current_epoch + 1
        log_dict['iter'] = trainer.inner_iter if self.by_epoch else trainer.current_iter
        cur_lr = trainer.lr_scheduler.get_lr()
        log_dict['lr'] = cur_lr[0] if isinstance(cur_lr, list) else cur_lr
        self._log_info(log_dict, trainer)

    def epoch_begin(self, trainer):
        self._pending.clear()
        trainer.logs.clear()

    def train_iter_end(self, trainer):
        for k, v in trainer.outputs.items():
            if torch.is_tensor(v):
                self._pending.setdefault(k, []).append(v)
            else:
                if k not in trainer.logs:
                    trainer.logs[k] = AverageMeter(k, ':.4e' if 'loss' in k else ':6.3f')
                trainer.logs[k].update(float(v))
        if self.by_epoch and self.every_n_inner_iters(trainer, self.interval):
            self.print_log(trainer)

    def train_epoch_end(self, trainer):
        self._flush(trainer)
        if self.reset_flag:
            trainer.logs.clear()
