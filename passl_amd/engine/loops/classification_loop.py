"""v2 classification loops — reference passl/engine/loops/classification_loop.py.

``ClassificationTrainingEpochLoop`` :36-101: ``train_one_step(batch) -> (out, loss_dict)`` with batch = [data, label]:
micro-batches of ``batch_size / accum_steps`` rows, ``out = model(data)``, ``loss_dict = train_loss_func(out, label)``
(each entry / accum_steps), backward, then grad_sync -> optimizer.step -> clear_grad -> lr_step(global_step) for
decay_unit 'step'; the logits of the micro-batches are concatenated for the train metric.

``ClassificationEvaluationLoop`` :104-262: ``run()`` evaluates ``trainer.eval_dataloader`` with the model in eval mode
(loss + metric per batch, averaged with the number of rows as weights; over several ranks the scores and labels of all
ranks are gathered and the samples DistributedBatchSampler repeated to fill the last batch are dropped), keeps
``latest_model_metric`` / ``best_model_metric`` (by the ``metric`` entry) and says whether this is a new best.

Differences by design (as engine/loops/loop.py): losses and metrics stay on the device until a line is printed or
the evaluation pass ends — one device->host transfer per print / per pass instead of one ``.item()`` per entry per
step; the weight-decay-free classifier of the linear-probe recipes is a single trainable arena, so the gradient
all-reduce is the overlapped GradReducer when the Engine built one (blocking grad_sync otherwise)."""
import collections
import logging
import time
from copy import deepcopy

import torch
import torch.distributed as dist

from ...core.sync_utils import collectives_active, grad_sync
from ...utils.misc import AverageMeter
from .loop import TrainingEpochLoop

logger = logging.getLogger('passl')


class ClassificationTrainingEpochLoop(TrainingEpochLoop):
    def __init__(self, trainer, epochs, max_train_step=None, val_loop=None):
        super().__init__(trainer, epochs, max_train_step=max_train_step, val_loop=val_loop)

    def forward_backward(self, batch):
        accum = getattr(self.trainer, 'accum_steps', 1)
        self.batch_size = batch[0].shape[0]
        assert self.batch_size % accum == 0, \
            'Bad accum_steps {} for batch size {}. This may be caused by two reasons: 1) the batch size setting is ' \
            'unreasonable and cannot be divisible, 2) drop_last in the sampler configuration is not set to ' \
            'True.'.format(accum, self.batch_size)
        step_size = self.batch_size // accum
        final_loss_dict = collections.defaultdict(float)
        final_out = []
        reducer = getattr(self.trainer, 'grad_reducer', None)
        for idx in range(accum):
            data = batch[0][idx * step_size:(idx + 1) * step_size]
            label = batch[1][idx * step_size:(idx + 1) * step_size]
            out = self.trainer.model(data)
            final_out.append(out)
            loss_dict = self.trainer.train_loss_func(out, label)
            for key in loss_dict:
                loss_dict[key] = loss_dict[key] / accum
                with torch.no_grad():
                    final_loss_dict[key] = final_loss_dict[key] + loss_dict[key].detach()
            if reducer is not None and idx == accum - 1:
                reducer.begin()            # overlap the all-reduce with the LAST micro-batch's backward
            loss_dict['loss'].backward()
        out = final_out[0] if len(final_out) == 1 else torch.cat(final_out, dim=0)
        return out, final_loss_dict

    def train_one_step(self, batch):
        out, loss_dict = self.forward_backward(batch)
        opt = self.trainer.optimizer
        if getattr(self.trainer, 'grad_reducer', None) is None:
            grad_sync([{'params': opt._parameter_list}])
        opt.step()
        opt.clear_grad()
        if getattr(self.trainer, 'lr_decay_unit', 'step') == 'step':
            sched = getattr(self.trainer, 'lr_scheduler', None)
            if sched is not None:
                sched.step(self.global_step)      # optimizer.lr_step(self.global_step), classification_loop.py:97-98
        # update_metric (loop.py:67-78): the train metric of the step joins the logged entries
        metric_func = getattr(self.trainer, 'train_metric_func', None)
        if metric_func is not None:
            with torch.no_grad():
                for k, v in metric_func(out.detach(), batch[1]).items():
                    loss_dict[k] = v
        return out, loss_dict


class ClassificationEvaluationLoop(object):
    def __init__(self, trainer):
        self.trainer = trainer
        self.best_model_metric = None
        self.best_model_to_save = False
        self.latest_model_metric = None
        self.time_info = {'reader_cost': AverageMeter('reader_cost'), 'batch_cost': AverageMeter('batch_cost')}

    def reset_state(self):
        self.best_model_to_save = False

    def update_best_model_metric_info(self):
        assert isinstance(self.latest_model_metric, dict)
        if 'metric' in self.latest_model_metric and (
                not self.best_model_metric or self.latest_model_metric['metric'] > self.best_model_metric['metric']):
            self.best_model_metric = deepcopy(self.latest_model_metric)
            self.best_model_to_save = True

    def run(self):
        assert self.trainer.mode in ['train', 'eval']
        assert self.trainer.validating is True
        self.reset_state()
        self.latest_model_metric = self.eval_one_dataset(self.trainer.eval_dataloader)
        if self.latest_model_metric is not None:
            self.update_best_model_metric_info()
        self.trainer.validating = False
        return self.latest_model_metric

    @torch.no_grad()
    def eval_one_dataset(self, eval_dataloader):
        tr = self.trainer
        tr.model.eval()
        world = dist.get_world_size() if collectives_active() else 1
        total_samples = len(eval_dataloader.dataset)
        accum_samples = 0
        sums, weights = {}, {}             # key -> device scalar sum of value * rows, rows
        tic = time.time()
        n_batches = len(eval_dataloader)
        for batch_idx, batch in enumerate(eval_dataloader):
            if batch_idx >= n_batches:
                break
            self.time_info['reader_cost'].update(time.time() - tic)
            data, label = batch[0], batch[1]
            batch_size = data.shape[0]
            out = tr.model(data)

            def add(d, rows):
                for k, v in d.items():
                    v = torch.as_tensor(v, dtype=torch.float32, device=data.device).reshape(()) * float(rows)
                    sums[k] = v if k not in sums else sums[k] + v
                    weights[k] = weights.get(k, 0) + rows
            if tr.eval_loss_func is not None:
                add(tr.eval_loss_func(out, label), batch_size)
            current_samples = batch_size * world
            accum_samples += current_samples
            if tr.eval_metric_func is not None:
                if world > 1:
                    logits = out['logits'] if isinstance(out, dict) else out
                    pred = torch.empty((world * batch_size,) + tuple(logits.shape[1:]), dtype=logits.dtype,
                                       device=logits.device)
                    labels = torch.empty((world * batch_size,) + tuple(label.shape[1:]), dtype=label.dtype,
                                         device=label.device)
                    dist.all_gather_into_tensor(pred, logits.contiguous())
                    dist.all_gather_into_tensor(labels, label.contiguous())
                    if accum_samples > total_samples:
                        # DistributedBatchSampler repeats samples to fill the last batch: drop the repeats
                        keep = total_samples + current_samples - accum_samples
                        pred, labels = pred[:keep], labels[:keep]
                        current_samples = keep
                    add(tr.eval_metric_func(pred, labels), current_samples)
                else:
                    add(tr.eval_metric_func(out, label), current_samples)
            self.time_info['batch_cost'].update(time.time() - tic)
            tic = time.time()
        if not sums:
            return None
        keys = list(sums)
        vals = torch.stack([sums[k] for k in keys]).cpu()          # the pass's only device->host transfer
        output_info = {k: float(vals[i]) / weights[k] for i, k in enumerate(keys)}
        logger.info('[Eval][Epoch {}][Avg]{}'.format(getattr(tr, 'cur_epoch_id', 0), ', '.join(
            '{}: {:.5f}'.format(k, v) for k, v in output_info.items())))
        if tr.eval_metric_func is None:
            return None
        return output_info
