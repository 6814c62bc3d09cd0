"""v2 ``Engine`` — reference passl/engine/engine.py:46-377: built from the v2 config schema
(``Global / Model / LRScheduler / Optimizer / DataLoader / DistributedStrategy``), owns model,
dataloader, lr scheduler, optimizer and the train loop named by ``Global.train_loop``
(``passl.engine.loops.<Name>``); ``Engine(config, mode='train').train()`` runs it.

A façade: everything it builds is what the v110 ``Trainer`` builds (same architectures through
``passl.models.build_model``, same flat-arena optimizers, same GradReducer) — no second code path and no
new kernels.  Kept from the reference: the attributes the loops read (``accum_steps``,
``print_batch_step``, ``save_interval``, ``lr_decay_unit``, ``optimizer``, ``lr_scheduler``, ``model``,
``train_dataloader``, ``config``, ``mode``, ``training``, ``output_dir``, ``model_name``), seed + rank
seeding, ``max_train_step``, data-parallel start-up broadcast.  Not carried over (outside the hot
path): FP16 GradScaler (an ``FP16:`` section — level O1 / O2 with its GradScaler, engine.py:177-210 — selects
reduced-precision compute, which here is bf16 with fp32 master weights and moments and needs no loss scaling;
``Global.compute_dtype`` overrides), EMA of the student weights, VisualDL, export, evaluation loops.
``runtime_info_hub`` (engine.py:346-349) carries epochs / max_steps / total_iterations to the models that read
them (MoCo-v3's momentum schedule).
"""
import inspect
import copy
import logging
import random

import numpy as np
import torch
import torch.distributed as dist

from ..core.sync_utils import GradReducer, collectives_active, param_sync
from ..hip import config as hip_config
from ..models import build_model
from ..solver.builder import LRSCHEDULERS, OPTIMIZERS
from ..utils.config import AttrDict
from ..utils.infohub import runtime_info_hub
from ..utils.registry import build_from_config
from . import loops
from .trainer import _init_distributed


class _ListBatchLoader(object):
    """Adapts the v110 loader (yields the views as a tuple) to the v2 batch convention
    ``[views, label]`` (contrastive_learning_loop.py:69 drops ``batch[-1]`` = the label)."""

    def __init__(self, inner):
        self.inner = inner

    def __len__(self):
        return len(self.inner)

    def __iter__(self):
        for views in self.inner:
            yield [list(views), None]


class Engine(object):
    def __init__(self, config, mode='train'):
        assert mode in ['train', 'eval', 'export']
        if mode != 'train':
            raise NotImplementedError('Engine(mode=%r): evaluation / export loops are outside the hot path' % mode)
        self.mode = mode
        self.config = config
        g = config['Global']
        self.print_batch_step = g.get('print_batch_step', 10)
        self.save_interval = g.get('save_interval', 1)
        self.accum_steps = g.get('accum_steps', 1)
        assert isinstance(self.accum_steps, int) and self.accum_steps > 0, \
            'accum_steps must be int dtype and greater than 0'
        self.max_train_step = g.get('max_train_step', None)
        self.checkpoint = g.get('checkpoint', None)           # resume prefix (engine.py:82)
        assert self.max_train_step is None or (isinstance(self.max_train_step, int) and self.max_train_step > 0), \
            'max_train_step must be int dtype and greater than 0'
        self.logger = logging.getLogger('passl')

        assert g['device'] in ['cpu', 'gpu']
        self.device = hip_config.set_device(g['device'])
        fp16 = config.get('FP16', None)
        if fp16 is not None:
            level = fp16.get('level', 'O2')
            assert level in ['O0', 'O1', 'O2']
            if not g.get('compute_dtype', None):
                hip_config.set_compute_dtype('fp32' if level == 'O0' else 'bf16')
            self.logger.info('FP16 level %s -> %s compute, fp32 master weights; GradScaler settings are not used '
                             '(bf16 keeps fp32\'s exponent range)', level, 'fp32' if level == 'O0' else 'bf16')
        if g.get('compute_dtype', None):
            hip_config.set_compute_dtype(g['compute_dtype'])
        rank, world = _init_distributed(self.device)
        g['distributed'], g['rank'], g['world_size'] = world != 1, rank, world

        seed = g.get('seed', False)
        if seed:
            assert isinstance(seed, int), "The 'seed' must be a integer!"
            seed += rank
            torch.manual_seed(seed)
            np.random.seed(seed)
            random.seed(seed)

        from ..datasets import build_dataloader
        dl = config['DataLoader']['Train']
        inner, _mix = build_dataloader(AttrDict(dataset=dl['dataset'], sampler=dl.get('sampler', {}),
                                                loader=dl.get('loader', {})), self.device)
        self.train_dataloader = _ListBatchLoader(inner)

        self.model = build_model(config['Model'])
        n_parameters = sum(p.numel() for p in self.model.parameters() if p.requires_grad)
        self.logger.info('Number of Parameters is {:.2f}M.'.format(n_parameters / 1e6))

        assert config.get('Optimizer', None) is not None, 'Optimizer must be defined in config.'
        opt_cfg = copy.deepcopy(dict(config['Optimizer']))
        self.lr_decay_unit = opt_cfg.pop('lr_decay_unit', None) or 'step'
        self.lr_scheduler = None
        sched_cfg = config.get('LRScheduler', None)
        if sched_cfg is not None:
            sched_cfg = AttrDict(copy.deepcopy(dict(sched_cfg)))
            self.lr_decay_unit = sched_cfg.pop('decay_unit', 'step')
            per_unit = len(self.train_dataloader) if self.lr_decay_unit == 'step' else 1
            if sched_cfg.name == 'CosineAnnealingDecay' and 'T_max' not in sched_cfg:
                sched_cfg.T_max = g['epochs'] * per_unit        # decay over the whole run
            # build_lr_scheduler (passl/scheduler/__init__.py:22-23) hands every v2 scheduler the run length
            accepted = inspect.signature(LRSCHEDULERS.get(sched_cfg.name).__init__).parameters
            if 'step_each_epoch' in accepted:
                sched_cfg.update({'epochs': g['epochs'], 'step_each_epoch': len(self.train_dataloader),
                                  'decay_unit': self.lr_decay_unit})
            self.lr_scheduler = build_from_config(sched_cfg, LRSCHEDULERS)
        name = opt_cfg.pop('name')
        params = list(self.model.parameters())
        lr = self.lr_scheduler if self.lr_scheduler is not None else opt_cfg.pop('learning_rate')
        kw = {'parameter_list' if 'Lars' in name else 'parameters': params}
        self.optimizer = OPTIMIZERS.get(name)(lr, **opt_cfg, **kw)

        if g.get('pretrained_model', None) is not None:
            assert isinstance(g['pretrained_model'], str), 'pretrained_model type is not available. Please use `string`.'
            self.model.load_pretrained(g['pretrained_model'], rank, g.get('finetune', False))

        self.grad_reducer = None
        if g['distributed'] or collectives_active():
            assert config.get('DistributedStrategy', None) is not None
            assert config['DistributedStrategy'].get('data_parallel', False) is True, \
                'If you want to use data parallel you should set data_parallel=True'
            arch = getattr(self.model, 'arch', self.model)
            param_sync(arch)
            self.grad_reducer = GradReducer(arch.arena_q if hasattr(arch, 'arena_q') else arch.arena,
                                            self.optimizer)

        train_loop_name = g.get('train_loop')
        self.train_loop = getattr(loops, train_loop_name)(self, epochs=g['epochs'],
                                                          max_train_step=self.max_train_step, val_loop=None)
        self.init_runtime_info_hub()

    def init_runtime_info_hub(self):
        runtime_info_hub.epochs = self.train_loop.epochs
        runtime_info_hub.max_steps = self.train_loop.max_steps
        runtime_info_hub.total_iterations = self.train_loop.global_step

    # ---- engine.py:319-347
    @property
    def cur_epoch_id(self):
        return self.train_loop.cur_epoch_id

    @property
    def global_step(self):
        return self.train_loop.global_step

    @property
    def epochs(self):
        return self.train_loop.epochs

    @property
    def model_name(self):
        return self.config['Model']['name']

    @property
    def output_dir(self):
        return self.config['Global']['output_dir']

    def train(self):
        assert self.mode == 'train'
        self.training = True
        self.model.train()
        self.train_loop.run()
