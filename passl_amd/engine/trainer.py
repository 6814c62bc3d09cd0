"""Trainer: build model / dataloader / lr / optimizer, wire the hook bus, run the iteration loop.

Mirrors the reference's ``Trainer`` (passl_v110/engine/trainer.py:72-337): same constructor
argument (the AttrDict config), same attributes that hooks read (model, optimizer, lr_scheduler,
outputs, logs, cfg, use_amp, current_iter, inner_iter, current_epoch, iters_per_epoch,
total_iters, epochs, output_dir, timestamp, logger, mode), same default hooks in the same order
(OptimizerHook, IterTimerHook, CheckpointHook, LogHook, LRSchedulerHook; stable sort by priority),
same loop (`train()`), and the model call contract
``model(*data, total_iters=..., current_iter=..., mixup_fn=...) -> dict with 'loss'``.

MI355X specifics: one process per GPU started by torchrun (RANK/LOCAL_RANK/WORLD_SIZE env);
``torch.distributed`` backend "nccl" (= RCCL over xGMI) on GPU, gloo on CPU; replicas are
initialised by a flat broadcast from rank 0 and gradients are averaged by the bucketed,
backward-overlapped GradReducer (passl_amd/core/sync_utils.py) — the role of
``fleet.distributed_model`` at trainer.py:172-183,218-219.
"""
import logging
import math
import os
import random
from collections import OrderedDict

import numpy as np
import torch
import torch.distributed as dist

from ..core.sync_utils import GradReducer, collectives_active, dp_forced, param_sync
from ..datasets import build_dataloader
from ..hip import config as hip_config
from ..utils.misc import AverageMeter
from ..hooks import Hook, build_hook
from ..modeling.architectures import build_model
from ..solver import build_lr_scheduler, build_lr_scheduler_simclr, build_optimizer


class IterLoader:
    """Endless view of a dataloader for the iteration-based loop: ``next()`` never raises, ``epoch``
    counts the passes that have been exhausted so far (the role of the loader wrapper the reference's
    ``Trainer.train`` draws from, passl_v110/engine/trainer.py:48-69)."""

    def __init__(self, dataloader, epoch=0):
        self._dataloader = dataloader
        self._epoch = epoch
        self._stream = self._batches()

    def _batches(self):
        while True:
            n = 0
            for batch in self._dataloader:
                n += 1
                yield batch
            if n == 0:
                raise RuntimeError('IterLoader: the dataloader yields no batch')
            self._epoch += 1

    @property
    def epoch(self):
        return self._epoch

    def __iter__(self):
        return self

    def __next__(self):
        return next(self._stream)

    def __len__(self):
        return len(self._dataloader)


def _init_distributed(device):
    world = int(os.environ.get('WORLD_SIZE', 1))
    if (world > 1 or (dp_forced() and 'RANK' in os.environ)) and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        # "nccl" IS RCCL on ROCm.  PASSL_DIST_BACKEND=gloo lets several ranks share ONE GPU (together with
        # PASSL_DEVICE_INDEX) to exercise the data-parallel path on a single-GPU box (tests/test_dp_gpu.py)
        backend = os.environ.get('PASSL_DIST_BACKEND') or ('nccl' if device.type == 'cuda' else 'gloo')
        # No ``device_id``: the RCCL communicator is then created at the first collective (the start-up broadcast of the
        # parameters, after the model exists) instead of here.  Measured on MI355X with a world-1 communicator and every
        # collective issued (profiles/r06_dp_overhead.txt): the same MoCo step takes 23.8 ms with the communicator
        # created late, 25.3 ms with it created here — merely EXISTING from this point on it slows the step by 8 %, with
        # no collective ever issued — against 23.2 ms without one.  PASSL_DIST_EAGER=1 restores the eager form.
        kw = {'device_id': device} if (device.type == 'cuda' and backend == 'nccl' and
                                       os.environ.get('PASSL_DIST_EAGER') == '1') else {}
        dist.init_process_group(backend=backend, **kw)
    return (dist.get_rank(), dist.get_world_size()) if dist.is_initialized() else (0, 1)


class Trainer:
    def __init__(self, cfg):
        self.logger = logging.getLogger('passl')
        self.cfg = cfg
        self.output_dir = cfg.output_dir
        self.log_interval = cfg.log_config.interval if 'log_config' in cfg else 10

        assert cfg['device'] in ['cpu', 'gpu']
        self.device = hip_config.set_device(cfg['device'])
        if cfg.get('compute_dtype', None):
            hip_config.set_compute_dtype(cfg['compute_dtype'])
        self.rank, self.world_size = _init_distributed(self.device)

        seed = cfg.get('seed', False)
        if seed:
            seed += self.rank                   # trainer.py:105-110
            torch.manual_seed(seed)
            np.random.seed(seed)
            random.seed(seed)

        self.start_epoch = 0
        self.current_epoch = 0
        self.current_iter = 0
        self.inner_iter = 0
        self.batch_id = 0
        self.global_steps = 0
        self.epochs = cfg.get('epochs', None)
        self.timestamp = cfg.get('timestamp', '')
        self.logs = OrderedDict()
        self.mixup_fn = None

        self.model = build_model(cfg.model)
        n_parameters = sum(p.numel() for p in self.model.parameters() if p.requires_grad)
        i = int(math.log(max(n_parameters, 1), 10) // 3)
        self.logger.info('Number of Parameters is {:.2f}{}.'.format(
            n_parameters / math.pow(1000, i), ['', 'K', 'M', 'B', 'T', 'Q'][i]))

        self.train_dataloader, self.mixup_fn = build_dataloader(cfg.dataloader.train, self.device)
        self.iters_per_epoch = len(self.train_dataloader)

        self.use_simclr_iters = cfg.get('use_simclr_iters', False)
        if self.use_simclr_iters:                 # trainer.py:157-163 (the x8 is the reference's)
            self.batch_size = cfg.dataloader.train.sampler.batch_size
            self.global_batch_size = cfg.global_batch_size
            self.lr_scheduler = build_lr_scheduler_simclr(
                cfg.lr_scheduler, self.iters_per_epoch, self.batch_size * 8, cfg.epochs,
                self.current_iter)
        else:
            self.lr_scheduler = build_lr_scheduler(
                cfg.lr_scheduler, self.iters_per_epoch, epochs=cfg.get('epochs', None),
                batch_size=int(cfg.dataloader.train.sampler.get('batch_size', 0) or 0) * self.world_size or None)
        self.optimizer = build_optimizer(cfg.optimizer, self.lr_scheduler, [self.model])

        self.use_amp = cfg.get('use_amp', False)
        self.scaler = None
        if self.use_amp:
            # paddle.amp O1/O2 (fp16 + GradScaler, trainer.py:186-215) has no counterpart here: the HIP path's mixed
            # precision is bf16 storage with fp32 accumulation and fp32 master weights, which needs no loss scaling
            hip_config.set_compute_dtype('bfloat16')
            self.logger.info('use_amp: mapped to compute_dtype=bfloat16 (fp32 master weights, no loss scaling; '
                             'AMP.level / scale_loss are ignored)')

        self.grad_reducer = None
        if (self.world_size > 1 or collectives_active()) and 'noreducer' not in os.environ.get('PASSL_DP_DIAG', ''):
            if 'noparamsync' not in os.environ.get('PASSL_DP_DIAG', ''):
                param_sync(self.model, src_rank=0)
            self.grad_reducer = GradReducer(self.model.arena_q, self.optimizer)

        self.hooks = []
        self.add_train_hooks()
        self.add_custom_hooks()
        self.hooks = sorted(self.hooks, key=lambda x: x.priority)
        self.step_graph = self._build_step_graph()

        if self.epochs:
            self.total_iters = self.epochs * self.iters_per_epoch
            self.by_epoch = True
        else:
            self.by_epoch = False
            self.total_iters = cfg.total_iters

    def add_train_hooks(self):
        for key, default in (('optimizer_config', 'OptimizerHook'), ('timer_config', 'IterTimerHook'),
                             ('checkpoint', 'CheckpointHook'), ('log_config', 'LogHook'),
                             ('lr_config', 'LRSchedulerHook')):
            c = self.cfg.get(key, None)
            self.add_hook(build_hook(c if c is not None else {'name': default}))

    def add_custom_hooks(self):
        custom_cfgs = self.cfg.get('custom_config', None)
        if custom_cfgs is None:
            return
        for custom_cfg in custom_cfgs:
            cfg_ = dict(custom_cfg)
            insert_index = cfg_.pop('insert_index', None)
            self.add_hook(build_hook(cfg_), insert_index)

    def add_hook(self, hook, insert_index=None):
        assert isinstance(hook, Hook)
        if insert_index is None:
            self.hooks.append(hook)
        elif isinstance(insert_index, int):
            self.hooks.insert(insert_index, hook)

    def call_hook(self, fn_name):
        for hook in self.hooks:
            getattr(hook, fn_name)(self)

    # ---- the step as a native launch plan (hip/replay.py) or one HIP graph (hip/graph.py) ---------------------
    def _build_step_graph(self):
        """Forward + OptimizerHook's clear_grad / backward / step recorded once and replayed.

        Default for models that opt in (``graph_safe``; MoCo): the NATIVE STEP PLAN (hip/replay.py) — the library's own
        launch list of the step, replayed from C on the recorded streams, forked branches and the key pipeline
        included; cfg ``step_plan: False`` or ``PASSL_PLAN=0`` keeps the eager step.  The HIP graph of round 3 stays
        available (cfg ``hip_graph: True`` / ``PASSL_GRAPH=1``; profiles/r03_graph_vs_eager.txt: bit-identical but no
        faster on this ROCm, and it cannot hold the forked branches).  Neither is used where a recorded step cannot be
        replayed faithfully: host tensors, MoCo's shuffle-BN (a fresh host-visible permutation every step), models
        that draw random numbers inside the step (MAE's masking noise), a custom OptimizerHook; the graph also not with
        collectives over gloo (host-staged) — a plan keeps collectives as live calls between its segments."""
        from ..hip.graph import StepGraph
        from ..hip.replay import StepPlan
        from ..hooks import OptimizerHook
        self._step_done = False
        opt_hooks = [h for h in self.hooks if isinstance(h, OptimizerHook)]
        want_graph = bool(self.cfg.get('hip_graph', os.environ.get('PASSL_GRAPH', '0') == '1'))
        want_plan = bool(self.cfg.get('step_plan', os.environ.get('PASSL_PLAN', '1') != '0')) and not want_graph
        ok = (self.device.type == 'cuda' and (want_graph or want_plan) and len(opt_hooks) == 1 and
              type(opt_hooks[0]) is OptimizerHook and hasattr(self.optimizer, 'push_hyper') and
              not getattr(self.model, 'shuffle_bn', False) and
              getattr(self.model, 'graph_safe', False))
        if ok and want_graph:
            ok = not dist.is_initialized() or dist.get_backend() == 'nccl'
        if not ok:
            return None
        hook = opt_hooks[0]

        def full_step(*data):
            self.outputs = None                 # see train_step
            self.outputs = self.model(*data, total_iters=self.total_iters, current_iter=self.current_iter,
                                      mixup_fn=self.mixup_fn)
            hook.optimize(self)
            return self.outputs
        replay_hooks = [self.model.on_graph_replay] if hasattr(self.model, 'on_graph_replay') else []
        if want_graph:
            return StepGraph(full_step, optimizers=[self.optimizer], replay_hooks=replay_hooks,
                             warmup=int(self.cfg.get('hip_graph_warmup', 3)))
        return StepPlan(full_step, optimizers=[self.optimizer], replay_hooks=replay_hooks,
                        warmup=int(self.cfg.get('step_plan_warmup', 3)))

    def train_step(self, data):
        """The body of one iteration between the ``train_iter_begin`` and ``train_iter_end`` hook calls: the
        model call of trainer.py:318-321, plus — when the step runs as a HIP graph — OptimizerHook's work, which
        the hook then skips."""
        # the previous step's outputs go first: the loss holds its autograd graph, and the nodes hold what the layers
        # parked on them (not only saved tensors) — kept across the next forward that is up to a third of a step's
        # activations twice (SimCLR R50 at 512 / GPU: 71 GB)
        self.outputs = None
        if self.step_graph is not None and self.mode == 'train':
            self.outputs = self.step_graph.run(*data)
            self._step_done = True
        else:
            self.outputs = self.model(*data, total_iters=self.total_iters, current_iter=self.current_iter,
                                      mixup_fn=self.mixup_fn)
        return self.outputs

    def train(self):
        self.mode = 'train'
        self.model.train()
        iter_loader = IterLoader(self.train_dataloader, self.current_epoch)
        self.call_hook('run_begin')
        while self.current_iter < self.total_iters:
            if self.current_iter % self.iters_per_epoch == 0:
                self.call_hook('train_epoch_begin')
            self.inner_iter = self.current_iter % self.iters_per_epoch
            self.current_iter += 1
            self.current_epoch = iter_loader.epoch
            data = next(iter_loader)
            self.call_hook('train_iter_begin')
            self.train_step(data)
            self.call_hook('train_iter_end')
            if self.current_iter % self.iters_per_epoch == 0:
                self.call_hook('train_epoch_end')
                self.current_epoch += 1
        self.call_hook('run_end')

    # ---- evaluation loop — reference trainer.py:339-417
    def val(self, **kargs):
        if not hasattr(self, 'val_dataloader'):
            self.val_dataloader, _mixup = build_dataloader(self.cfg.dataloader.val, self.device)
        self.logger.info('start evaluate on epoch {} ..'.format(self.current_epoch + 1))
        rank, world_size = self.rank, self.world_size
        model = self.model
        total_samples = len(self.val_dataloader.dataset)
        self.logger.info('Evaluate total samples {}'.format(total_samples))
        accum_samples = 0
        self.model.eval()
        outs = OrderedDict()
        for data in self.val_dataloader:
            batch_size = data.shape[0] if torch.is_tensor(data) else data[0].shape[0]
            labels = data[-1]
            pred = model(*data, mode='test')
            current_samples = batch_size * world_size
            accum_samples += current_samples
            if world_size > 1:
                pred_all = torch.empty((world_size * pred.shape[0],) + tuple(pred.shape[1:]), dtype=pred.dtype,
                                       device=pred.device)
                dist.all_gather_into_tensor(pred_all, pred.contiguous())
                lab_all = torch.empty(world_size * labels.shape[0], dtype=labels.dtype, device=labels.device)
                dist.all_gather_into_tensor(lab_all, labels.contiguous().view(-1))
                pred, labels = pred_all, lab_all
                if accum_samples > total_samples:
                    keep = total_samples + current_samples - accum_samples
                    self.logger.info('total samples {} {} {}'.format(total_samples, accum_samples, keep))
                    pred, labels = pred[:keep], labels[:keep]
                    current_samples = keep
            res = self.val_dataloader.dataset.evaluate(pred, labels, **kargs)
            for k, v in res.items():
                if k not in outs:
                    outs[k] = AverageMeter(k, ':6.3f')
                outs[k].update(float(v), current_samples)
        log_items = ['{} ({:6.3f})'.format(m.name, m.avg) for m in outs.values()]
        self.logger.info(f'Validate Epoch [{self.current_epoch + 1}] ' + ', '.join(log_items))
        self.val_results = OrderedDict((k, m.avg) for k, m in outs.items())
        self.model.train()
        return self.val_results

    # ---- checkpoint plumbing (scope row §8f-3) — reference trainer.py:419-444
    def resume(self, checkpoint_path):
        from ..utils.checkpoint import load_pickle, to_tensors
        checkpoint = load_pickle(checkpoint_path)
        if checkpoint.get('epoch', None) is not None:
            self.start_epoch = checkpoint['epoch']
            self.current_epoch = checkpoint['epoch']
            # as in the reference (trainer.py:424): the iteration counter restarts one epoch early
            self.current_iter = (self.start_epoch - 1) * self.iters_per_epoch
        self.load_numpy_state(checkpoint['state_dict'])
        self.optimizer.set_state_dict(to_tensors(checkpoint['optimizer']))
        self.lr_scheduler.set_state_dict(checkpoint['lr_scheduler'])
        self.logger.info('Resume training from {} success!'.format(checkpoint_path))

    def load(self, weight_path, export=False):
        from ..utils.checkpoint import load_pickle
        state_dict = load_pickle(weight_path)
        if 'state_dict' in state_dict:
            state_dict = state_dict['state_dict']
        if export:
            raise NotImplementedError('export (paddle.jit inference model) is outside the hot path')
        self.load_numpy_state(state_dict)

    def load_numpy_state(self, sd):
        from ..utils.checkpoint import load_lenient
        load_lenient(self.model, sd, self.logger, what='checkpoint')
