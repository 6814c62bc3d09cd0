"""TrainingEpochLoop — the epoch/iteration driver of the v2 engine, reference
passl/engine/loops/loop.py:141-311 (``run`` -> ``train_one_epoch`` -> ``train_one_step``; global step
counter, ``max_train_step`` early stop, timers reset after 5 warm-up iterations, ``ips`` =
batch_size * world_size / batch_cost in the log line, lr stepping per epoch when
``lr_decay_unit == 'epoch'``, checkpoint every ``save_interval`` epochs; with a validation loop:
``_should_check_val`` (loop.py:51-64) after an epoch / every ``eval_interval`` units, then a checkpoint that also
carries the metric, plus a ``best`` copy when the ``metric`` entry improved).

Differences by design: losses are kept as device tensors and only converted when a line is printed
(the reference calls ``.item()`` every iteration = one device->host sync per step, loop.py:85);
``max_train_step`` ends the loop by returning instead of ``exit(0)``."""
import datetime
import logging
import os
import time

import torch

from ...utils.misc import AverageMeter

logger = logging.getLogger('passl')


class TrainingEpochLoop(object):
    def __init__(self, trainer, epochs, max_train_step=None, val_loop=None):
        self.trainer = trainer
        self.start_eopch = 0                      # (sic) the reference's attribute name
        self.epochs = epochs
        self.cur_epoch_id = 0
        self.global_step = 0
        self.max_train_step = max_train_step
        self.val_loop = val_loop
        self.time_info = {'reader_cost': AverageMeter('reader_cost'), 'batch_cost': AverageMeter('batch_cost')}
        self.output_info = {}
        self._pending = []

    @property
    def max_steps(self):
        return self.epochs * len(self.trainer.train_dataloader)

    # ---- loop.py:207-252
    def run(self):
        assert self.trainer.mode == 'train' and self.trainer.training is True
        self.resume()
        self.total_batch_idx = len(self.trainer.train_dataloader)
        for epoch_id in range(self.start_eopch + 1, self.epochs + 1):
            self.cur_epoch_id = epoch_id
            if self.val_loop is not None:
                # loop.py:200-202 (reset_state, every epoch): a `best` found by an earlier epoch's evaluation must
                # not tag THIS epoch's (unevaluated) weights when eval_interval > 1 or eval_unit is 'step'
                self.val_loop.best_model_to_save = False
            stop = self.train_one_epoch()
            if self.trainer.lr_decay_unit == 'epoch' and self.trainer.lr_scheduler is not None:
                self.trainer.lr_scheduler.step(self.cur_epoch_id)
            self._flush()
            logger.info('[Train][Epoch {}/{}][Avg]{}'.format(epoch_id, self.epochs, ', '.join(
                '{}: {:.5f}'.format(k, m.avg) for k, m in self.output_info.items())))
            self.output_info.clear()
            if stop:
                break
            to_save = epoch_id % self.trainer.save_interval == 0 or epoch_id == self.epochs
            if self._should_check_val():
                self.trainer.validating = True
                self.val_loop.run()
                self.trainer.training = True
                self.trainer.model.train()
                to_save = True
                best = self.val_loop.best_model_metric
                if best is not None and 'metric' in best:
                    logger.info('[Eval][Epoch {}][best metric: {}]'.format(self.cur_epoch_id, best['metric']))
            if to_save or (self.val_loop is not None and self.val_loop.best_model_to_save):
                self.save_checkpoint()
        self.trainer.training = False

    def _should_check_val(self):
        """loop.py:51-64."""
        if self.val_loop is None:
            return False
        g = self.trainer.config['Global']
        if not g.get('eval_during_train', False):
            return False
        interval = g.get('eval_interval', 1)
        if g.get('eval_unit', 'epoch') == 'epoch':
            return self.cur_epoch_id % interval == 0
        return self.global_step % interval == 0

    # ---- loop.py:255-308
    def train_one_epoch(self):
        self.trainer.model.train()
        tic = time.time()
        for batch_idx, batch in enumerate(self.trainer.train_dataloader):
            self.cur_batch_idx = batch_idx
            if self.max_train_step is not None and self.global_step >= self.max_train_step:
                logger.info('global_step({}) >= max_train_step({}), training stops early.'.format(
                    self.global_step, self.max_train_step))
                return True
            if batch_idx >= self.total_batch_idx:
                break
            if batch_idx == 5:
                for m in self.time_info.values():
                    m.reset()
            self.time_info['reader_cost'].update(time.time() - tic)
            self.global_step += 1
            _out, loss_dict = self.train_one_step(batch)
            self.time_info['batch_cost'].update(time.time() - tic)
            self._pending.append(loss_dict)
            if batch_idx % self.trainer.print_batch_step == 0:
                self.log_info()
            tic = time.time()
        return False

    def train_one_step(self, batch):
        raise NotImplementedError

    def _flush(self):
        """One device->host transfer for everything queued since the last print."""
        if not self._pending:
            return
        keys = list(self._pending[0])
        vals = torch.stack([torch.stack([torch.as_tensor(d[k]).detach().reshape(()).float() for k in keys])
                            for d in self._pending]).cpu()
        for j, k in enumerate(keys):
            m = self.output_info.setdefault(k, AverageMeter(k))
            for i in range(vals.shape[0]):
                m.update(float(vals[i, j]), self.batch_size)
        self._pending = []

    def log_info(self):
        self._flush()
        world = self.trainer.config['Global'].get('world_size', 1)
        cost = max(self.time_info['batch_cost'].avg, 1e-9)
        eta = ((self.epochs - self.cur_epoch_id + 1) * self.total_batch_idx - self.cur_batch_idx) * cost
        logger.info('[Train][Epoch {}/{}][Iter: {}/{}] lr: {:.6f}, {}, {}, ips: {:.5f} images/sec, eta: {}'.format(
            self.cur_epoch_id, self.epochs, self.cur_batch_idx, self.total_batch_idx,
            self.trainer.optimizer.get_lr(),
            ', '.join('{}: {:.5f}'.format(k, m.avg) for k, m in self.output_info.items()),
            ', '.join('{}: {:.5f}'.format(k, m.avg) for k, m in self.time_info.items()),
            self.batch_size * world / cost, datetime.timedelta(seconds=int(eta))))

    # ---- loop.py:317-340 over passl/utils/io.py:115-170
    def save_checkpoint(self):
        """<output_dir>/<model_name>/epoch_N.{pdparams,pdopt,pdstates} and the same three under ``latest``:
        model (``Model.save``), optimizer + lr-scheduler state, and {epoch, global_step, timestamp} — the set
        ``resume`` needs to continue a run."""
        import pickle
        from ...utils.checkpoint import to_numpy
        rank = self.trainer.config['Global'].get('rank', 0)
        model_dir = os.path.join(self.trainer.output_dir, self.trainer.model_name)
        prefixes = [os.path.join(model_dir, 'epoch_{}'.format(self.cur_epoch_id)), os.path.join(model_dir, 'latest')]
        metric_info = {}
        if self.val_loop is not None:
            if self.val_loop.best_model_to_save:
                prefixes.append(os.path.join(model_dir, 'best'))          # io.save_checkpoint(is_best=True)
                metric_info = dict(self.val_loop.best_model_metric)
            elif self.val_loop.latest_model_metric is not None:
                metric_info = dict(self.val_loop.latest_model_metric)
        for prefix in prefixes:
            self.trainer.model.save(prefix, rank=rank)
        if rank != 0:
            return
        os.makedirs(model_dir, exist_ok=True)
        opt_state = to_numpy(self.trainer.optimizer.state_dict())
        metric_info.update({'epoch': self.cur_epoch_id, 'global_step': self.global_step,
                            'timestamp': time.strftime('%Y-%m-%d %H:%M:%S', time.localtime(time.time()))})
        for prefix in prefixes:
            with open(prefix + '.pdopt', 'wb') as f:
                pickle.dump(opt_state, f, protocol=2)
            with open(prefix + '.pdstates', 'wb') as f:
                pickle.dump(metric_info, f, protocol=2)
        logger.info('Already save epoch_{} model in {}'.format(self.cur_epoch_id, model_dir))
        self._prune_checkpoints(model_dir)

    def _prune_checkpoints(self, model_dir):
        """passl/utils/io.py:172-201: ``Global.max_num_latest_checkpoint`` = N >= 0 keeps the N most recent ``epoch_*``
        checkpoints (ordered by the timestamp stored in their .pdstates; 0 — the task yamls' value — keeps none:
        only ``latest`` / ``best`` survive); absent or negative keeps all.  As in the reference the files removed are
        <prefix>.{pdparams,pdopt,pdstates}; the encoder-only exports a model's ``save`` writes next to them stay."""
        import glob
        from ...utils.checkpoint import load_pickle
        keep = self.trainer.config['Global'].get('max_num_latest_checkpoint', -1)
        if keep is None or keep < 0:
            return
        # (the reference matches 'best' / 'latest' anywhere in the PATH and keys the files by a one-second timestamp
        # string: an output_dir called 'latest_run' switches pruning off and two checkpoints of the same second
        # collide.  Here: the file's own name decides, and the order is (epoch in the name, stored timestamp, mtime).)
        entries = []
        for path in glob.glob(os.path.join(model_dir, '*.pdstates')):
            stem = os.path.basename(path)[:-len('.pdstates')]
            if stem in ('best', 'latest') or not stem.startswith('epoch_'):
                continue
            try:
                epoch = int(stem[len('epoch_'):])
            except ValueError:
                epoch = -1
            try:
                ts = load_pickle(path).get('timestamp', '')
            except Exception:
                ts = ''
            entries.append(((epoch, ts, os.path.getmtime(path)), path[:-len('.pdstates')]))
        entries.sort()
        for _key, prefix in (entries[:-keep] if keep > 0 else entries):
            for ext in ('.pdparams', '.pdopt', '.pdstates'):
                try:
                    os.remove(prefix + ext)
                except OSError:
                    pass

    # ---- loop.py:358-375 over passl/utils/io.py:52-96
    def resume(self):
        """``Global.checkpoint`` (a path prefix without extension): model, optimizer / lr-scheduler state and the
        epoch / step counters of a previous run; training continues with epoch ``epoch + 1``."""
        ckpt = getattr(self.trainer, 'checkpoint', None)
        if ckpt is None:
            return
        from ...utils.checkpoint import load_pickle
        assert isinstance(ckpt, str), 'checkpoint type is not available. Please use `string`.'
        rank = self.trainer.config['Global'].get('rank', 0)
        self.trainer.model.load_pretrained(ckpt, rank=rank, finetune=False)
        opt_path = ckpt + '.pdopt'
        assert os.path.exists(opt_path), 'Optimizer checkpoint path {} does not exists.'.format(opt_path)
        opt_state = load_pickle(opt_path)
        self.trainer.optimizer.set_state_dict(opt_state)
        if os.path.exists(ckpt + '.pdstates'):
            metric_info = load_pickle(ckpt + '.pdstates')
            if self.val_loop is not None and 'metric' in metric_info:
                self.val_loop.best_model_metric = dict(metric_info)       # loop.py:370-371: the bar a new best must pass
            if 'global_step' in metric_info:
                self.global_step = int(metric_info['global_step'])
            if 'epoch' in metric_info:
                self.start_eopch = int(metric_info['epoch'])
        logger.info('Finish load checkpoint from {}'.format(ckpt))
