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
``Global.compute_dtype`` overrides), EMA of the student weights, VisualDL, export.
``task_type: Classification`` (the linear-probe recipes of tasks/ssl/{simsiam,mocov3}): ``Loss`` / ``Metric`` blocks
(engine.py:142-177), the ``DataLoader.Eval`` loader and ``Global.validate_loop`` when ``eval_during_train`` or
``mode='eval'`` (engine.py:135-141, 301-306), ``Engine.eval()`` (engine.py:361-367).
``runtime_info_hub`` (engine.py:346-349) carries epochs / max_steps / total_iterations to the models that read
them (MoCo-v3's momentum schedule).
"""
import inspect
import copy
import os
import logging
import random

import numpy as np
import torch
import torch.distributed as dist

from ..core.sync_utils import GradReducer, ReducerGroup, collectives_active, param_sync
from ..hip import config as hip_config
from ..models import build_model
from ..solver.builder import LRSCHEDULERS, OPTIMIZERS
from ..utils.config import AttrDict
from ..utils.infohub import runtime_info_hub
from ..utils.registry import build_from_config
from . import loops
from .trainer import _init_distributed


def _plain(x):
    """AttrDict / list config nodes -> plain dicts and lists (the loss / metric builders pop from them)."""
    if isinstance(x, dict):
        return {k: _plain(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_plain(v) for v in x]
    return x


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


class _LabeledBatchLoader(object):
    """(image, label) batches of the synthetic labeled source as the v2 classification loops take them:
    ``batch[0]`` = data, ``batch[1]`` = label (classification_loop.py:52-53); ``dataset`` for the sample count."""

    def __init__(self, inner):
        self.inner = inner
        self.dataset = inner.dataset

    def __len__(self):
        return len(self.inner)

    def __iter__(self):
        for img, label in self.inner:
            yield [img, label]


class _Schedulers(object):
    """Several schedules stepped as one (the default schedule + those of parameter groups:
    Optimizer.lr_step, passl/optimizer/optimizer.py:216-222)."""

    def __init__(self, scheds):
        self.scheds = list(scheds)

    def step(self, epoch=None):
        for s_ in self.scheds:
            s_.step(epoch)

    def __call__(self):
        return self.scheds[0]()

    def __getattr__(self, name):
        # (only called for names the instance does not have: `scheds` itself is missing while copy / pickle rebuild
        # the object — forwarding then would recurse until RecursionError)
        if name == 'scheds' or name.startswith('__'):
            raise AttributeError(name)
        return getattr(self.scheds[0], name)

    def state_dict(self):
        return {'sched_%d' % i: s_.state_dict() for i, s_ in enumerate(self.scheds)}

    def set_state_dict(self, sd):
        for i, s_ in enumerate(self.scheds):
            s_.set_state_dict(sd['sched_%d' % i])


class OptimizerGroup(object):
    """One flat-arena optimizer per parameter group (each group must cover whole arenas: the model puts the
    groups of its task yaml into separate arenas), behind the optimizer API the loops use."""

    def __init__(self, klass, key, groups, default_lr, common, build_scheduler):
        self.optimizers, self.names, self.schedulers = [], [], []
        for c, plist in groups:
            kw = dict(common)
            lr = default_lr
            if isinstance(c.get('lr', None), dict):
                lr = build_scheduler(c['lr'])
                self.schedulers.append(lr)
            elif c.get('lr', None) is not None:
                lr = float(c['lr'])
            for k in ('weight_decay', 'momentum'):
                if k in c:
                    kw[k] = c[k]
            self.optimizers.append(klass(lr, **kw, **{key: plist}))
            self.names.append(c['name'])
        self._parameter_list = [p for o in self.optimizers for p in o._parameter_list]

    @property
    def grad_scale(self):
        return self.optimizers[0].grad_scale

    @grad_scale.setter
    def grad_scale(self, v):
        for o in self.optimizers:
            o.grad_scale = v

    def get_lr(self, group_id=0):
        return self.optimizers[group_id].get_lr()

    def step(self):
        for o in self.optimizers:
            o.step()

    def clear_grad(self, set_to_zero=True):
        for o in self.optimizers:
            o.clear_grad()

    clear_gradients = clear_grad

    def state_dict(self):
        return {'group_%d' % i: o.state_dict() for i, o in enumerate(self.optimizers)}

    def set_state_dict(self, sd):
        for i, o in enumerate(self.optimizers):
            o.set_state_dict(sd['group_%d' % i])


class Engine(object):
    def __init__(self, config, mode='train'):
        assert mode in ['train', 'eval', 'export']
        if mode == 'export':
            raise NotImplementedError('Engine(mode=%r): export (paddle.jit inference model) is outside the hot path' % mode)
        self.mode = mode
        self.training = False
        self.validating = False
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
        classification = g.get('task_type', None) == 'Classification' or \
            str(g.get('train_loop', '')).startswith('Classification')
        eval_wanted = mode == 'eval' or (mode == 'train' and g.get('eval_during_train', False))

        def loader(block):
            inner, _mix = build_dataloader(AttrDict(dataset=block['dataset'], sampler=block.get('sampler', {}),
                                                    loader=block.get('loader', {})), self.device)
            return _LabeledBatchLoader(inner) if classification else _ListBatchLoader(inner)
        self.train_dataloader = loader(config['DataLoader']['Train']) if mode == 'train' else None
        self.eval_dataloader = None
        if eval_wanted and config['DataLoader'].get('Eval', None) is not None:
            self.eval_dataloader = loader(config['DataLoader']['Eval'])

        # build loss / metric (engine.py:142-177)
        from ..loss import build_loss
        from ..metric import build_metrics
        self.train_loss_func = self.eval_loss_func = self.train_metric_func = self.eval_metric_func = None
        loss_cfg, metric_cfg = config.get('Loss', None), config.get('Metric', None)
        if loss_cfg is not None:
            if mode == 'train' and loss_cfg.get('Train', None) is not None:
                self.train_loss_func = build_loss(_plain(loss_cfg['Train']))
            if eval_wanted and loss_cfg.get('Eval', None) is not None:
                self.eval_loss_func = build_loss(_plain(loss_cfg['Eval']))
        if metric_cfg is not None:
            if mode == 'train' and metric_cfg.get('Train', None) is not None:
                self.train_metric_func = build_metrics(_plain(metric_cfg['Train']))
            if eval_wanted and metric_cfg.get('Eval', None) is not None:
                self.eval_metric_func = build_metrics(_plain(metric_cfg['Eval']))

        self.model = build_model(config['Model'])
        n_parameters = sum(p.numel() for p in self.model.parameters() if p.requires_grad)
        self.logger.info('Number of Parameters is {:.2f}M.'.format(n_parameters / 1e6))

        self.optimizer = self.lr_scheduler = None
        self.lr_decay_unit = 'step'
        if mode == 'train':
            self._build_optimizer(config, g)

        if g.get('pretrained_model', None) is not None:
            assert isinstance(g['pretrained_model'], str), 'pretrained_model type is not available. Please use `string`.'
            self.model.load_pretrained(g['pretrained_model'], rank, g.get('finetune', False))

        self.grad_reducer = None
        if (g['distributed'] or collectives_active()) and mode == 'train':
            assert config.get('DistributedStrategy', None) is not None
            assert config['DistributedStrategy'].get('data_parallel', False) is True, \
                'If you want to use data parallel you should set data_parallel=True'
            arch = getattr(self.model, 'arch', self.model)
            param_sync(arch)
            arenas = arch.trainable_arenas() if hasattr(arch, 'trainable_arenas') else None
            if arenas is not None and len(arenas) > 1 and os.environ.get('PASSL_DP_BLOCKING_GROUPS') == '1':
                # several trainable arenas (parameter groups): the loop's blocking grad_sync reduces each flat
                # gradient buffer after backward (the reference's behaviour, sync_utils.py:18-43; kept as a switch)
                self.grad_reducer = None
            elif arenas is not None and len(arenas) > 1:
                # one overlapped reducer per arena: the buckets are all-reduced from inside the backward pass
                self.grad_reducer = ReducerGroup(arenas, self.optimizer)
            else:
                self.grad_reducer = GradReducer(arch.arena_q if hasattr(arch, 'arena_q') else arch.arena,
                                                self.optimizer)

        # build train_loop and eval_loop (engine.py:301-316)
        self.validate_loop = None
        if g.get('validate_loop', None) is not None and self.eval_dataloader is not None:
            self.validate_loop = getattr(loops, g['validate_loop'])(self)
        self.train_loop = None
        if mode == 'train':
            self.train_loop = getattr(loops, g.get('train_loop'))(self, epochs=g['epochs'],
                                                                  max_train_step=self.max_train_step,
                                                                  val_loop=self.validate_loop)
            self.init_runtime_info_hub()

    def _build_optimizer(self, config, g):
        """engine.py:212-236 over build_lr_scheduler / build_optimizer."""
        assert config.get('Optimizer', None) is not None, 'Optimizer must be defined in config.'
        opt_cfg = copy.deepcopy(dict(config['Optimizer']))
        self.lr_decay_unit = opt_cfg.pop('lr_decay_unit', None) or 'step'
        self.lr_scheduler = None
        sched_cfg = config.get('LRScheduler', None)
        if sched_cfg is not None:
            self.lr_decay_unit = sched_cfg.get('decay_unit', 'step')
            self.lr_scheduler = self._build_scheduler(dict(sched_cfg), g)
        name = opt_cfg.pop('name')
        # build_optimizer (passl/optimizer/__init__.py:124-212): `lr` inside the Optimizer block is the default
        # schedule (a float or a scheduler config, its decay_unit reset to lr_decay_unit); `param_groups` split the
        # trainable parameters by name (re.match(group name, parameter name), first match wins, the rest is
        # 'default') and may carry their own `lr`
        lr_cfg = opt_cfg.pop('lr', None)
        if isinstance(lr_cfg, dict):
            self.lr_scheduler = self._build_scheduler(dict(lr_cfg, decay_unit=self.lr_decay_unit), g)
        groups_cfg = opt_cfg.pop('param_groups', None)
        if self.lr_scheduler is not None:
            lr = self.lr_scheduler
        elif isinstance(lr_cfg, (int, float)):
            lr = float(lr_cfg)
        else:
            lr = opt_cfg.pop('learning_rate')
        klass = OPTIMIZERS.get(name)
        key = 'parameter_list' if 'Lars' in name else 'parameters'
        # keys of the reference's Optimizer block (passl/optimizer/__init__.py:124-212) that the flat-arena optimizers
        # do not implement: say so by name instead of a TypeError from the constructor.  `tensor_fusion` asks Paddle to
        # fuse parameter storage — the arena IS fused storage, nothing to do.
        opt_cfg.pop('tensor_fusion', None)
        if opt_cfg.get('grad_clip', None) is None:
            opt_cfg.pop('grad_clip', None)
        for k in ('grad_clip', 'no_weight_decay_name', 'layer_decay'):
            if opt_cfg.get(k, None) not in (None, [], ''):
                raise NotImplementedError('Optimizer.%s is not built on the HIP path (pre-training recipes of '
                                          'tasks/ssl do not use it)' % k)
            opt_cfg.pop(k, None)
        if groups_cfg:
            self.optimizer = OptimizerGroup(klass, key, self._group_params(groups_cfg), lr, opt_cfg,
                                            lambda c: self._build_scheduler(dict(c, decay_unit=self.lr_decay_unit), g))
            if self.optimizer.schedulers:
                self.lr_scheduler = _Schedulers([s_ for s_ in [self.lr_scheduler] if s_ is not None] +
                                                self.optimizer.schedulers)
        else:
            self.optimizer = klass(lr, **opt_cfg, **{key: list(self.model.parameters())})

    def _build_scheduler(self, sched_cfg, g):
        """build_lr_scheduler (passl/scheduler/__init__.py:22-36): every v2 scheduler is handed the run length."""
        sched_cfg = AttrDict(copy.deepcopy(dict(sched_cfg)))
        unit = sched_cfg.pop('decay_unit', self.lr_decay_unit)
        per_unit = len(self.train_dataloader) if unit == 'step' else 1
        if sched_cfg.name == 'CosineAnnealingDecay' and 'T_max' not in sched_cfg:
            sched_cfg.T_max = g['epochs'] * per_unit        # decay over the whole run
        accepted = inspect.signature(LRSCHEDULERS.get(sched_cfg.name).__init__).parameters
        if 'step_each_epoch' in accepted:
            sched_cfg.update({'epochs': g['epochs'], 'step_each_epoch': len(self.train_dataloader), 'decay_unit': unit})
        return build_from_config(sched_cfg, LRSCHEDULERS)

    def _group_params(self, groups_cfg):
        """group_params (passl/optimizer/__init__.py:68-113) -> [(group config, [parameters])] in config order,
        'default' last."""
        import re
        groups = [(dict(c), []) for c in groups_cfg]
        default = []
        for pname, p in self.model.named_parameters():
            if not p.requires_grad:
                continue
            for c, plist in groups:
                if re.compile(c.get('regular_exp', c['name'])).match(pname):
                    plist.append(p)
                    break
            else:
                default.append(p)
        if default:
            groups.append(({'name': 'default'}, default))
        for c, plist in groups:
            self.logger.info('%s-params length: %d', c['name'], len(plist))
        return groups

    def init_runtime_info_hub(self):
        runtime_info_hub.epochs = self.train_loop.epochs
        runtime_info_hub.max_steps = self.train_loop.max_steps
        runtime_info_hub.total_iterations = self.train_loop.global_step

    # ---- engine.py:319-347
    @property
    def cur_epoch_id(self):
        return self.train_loop.cur_epoch_id if self.train_loop is not None else 0

    @property
    def global_step(self):
        return self.train_loop.global_step if self.train_loop is not None else 0

    @property
    def epochs(self):
        return self.train_loop.epochs if self.train_loop is not None else 0

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

    # ---- engine.py:361-367
    def eval(self):
        assert self.mode in ['train', 'eval']
        if self.validate_loop is None:
            raise RuntimeError('Engine.eval() needs Global.validate_loop and a DataLoader.Eval block')
        self.model.eval()
        self.validating = True
        eval_result = self.validate_loop.run()
        self.model.train()
        return eval_result
