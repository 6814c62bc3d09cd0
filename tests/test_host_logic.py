"""Host-side mirror of the reference interface, exercised on CPU: registries, config +
overrides, lr schedule, hook bus order, Trainer loop, model construction / state_dict layout.
(The HIP layers are built but not run here; running them needs the GPU tests.)"""
import math
import os

import numpy as np
import pytest
import torch

from passl_amd.hip import config as hip_config
from passl_amd.utils.config import AttrDict, get_config, override_config
from passl_amd.utils.registry import Registry, build_from_config

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_CFG = '/root/reference/configs/moco/moco_v2_r50.yaml'
OUR_CFG = os.path.join(ROOT, 'configs/moco/moco_v2_r50_synthetic.yaml')


def test_registry_contract():
    R = Registry('T')

    @R.register()
    class A(object):
        def __init__(self, x=1):
            self.x = x
    R.register(dict, name='D')
    assert R.get('A') is A and R.get('D') is dict
    with pytest.raises(AssertionError):
        R.register(A)                         # duplicate name
    with pytest.raises(KeyError):
        R.get('missing')
    assert build_from_config({'name': 'A', 'x': 3}, R).x == 3
    assert build_from_config({'x': 5}, R, default_args={'name': 'A'}).x == 5
    with pytest.raises(KeyError):
        build_from_config({'x': 5}, R)
    with pytest.raises(TypeError):
        build_from_config({'name': 'A', 'bogus': 1}, R)      # constructor error is re-raised
    with pytest.raises(TypeError):
        build_from_config(['name'], R)


def test_config_overrides():
    cfg = get_config(OUR_CFG, ['optimizer.weight_decay=0.5', 'epochs=3',
                               'dataloader.train.sampler.batch_size=64'])
    assert cfg.optimizer.weight_decay == 0.5 and cfg.epochs == 3
    assert cfg.dataloader.train.sampler.batch_size == 64
    assert isinstance(cfg.model.backbone, AttrDict) and cfg.model.backbone.depth == 50
    with pytest.raises(AssertionError):
        override_config(cfg, ['model.nonexistent=1'])
    with pytest.raises(AssertionError):
        get_config('/nonexistent.yaml')


@pytest.mark.skipif(not os.path.exists(REF_CFG), reason='reference tree not present')
def test_reference_moco_config_loads_and_builds_unchanged():
    hip_config.set_device('cpu')
    from passl_amd.modeling import build_model
    cfg = get_config(REF_CFG, ['dataloader.train.dataset.name=SyntheticTwoView'])
    assert cfg.model.head.temperature == 0.2 and cfg.lr_scheduler.T_max == 200
    assert cfg.dataloader.train.dataset.view_trans1[5].scale == '1.0/255.0'   # literal_eval keeps it
    model = build_model(cfg.model)
    assert model.K == 65536 and model.m == 0.999 and model.head.temperature == 0.2
    assert sum(p.numel() for p in model.parameters() if p.requires_grad) == 27966656


def test_model_state_layout_matches_reference_naming():
    """state_dict keys/shapes = SURVEY appendix A (and the oracle's, which is checked against the
    reference's own state_dict order in oracle/ref_runner.load_oracle_state)."""
    hip_config.set_device('cpu')
    from oracle.resnet50 import init_encoder_state
    from passl_amd.modeling import build_model
    import moco_util as U
    cfg = dict(U.MODEL_CFG)
    cfg.update(K=256)
    model = build_model(cfg)
    ost = init_encoder_state(torch.Generator().manual_seed(0))
    qsd = model.encoder_q.state_dict()
    assert list(qsd.keys()) == list(ost.keys())
    for k in ost:
        assert tuple(qsd[k].shape) == tuple(ost[k].shape), k
    full = model.state_dict()
    assert 'queue' in full and 'queue_ptr' in full and full['queue'].shape == (128, 256)
    assert 'backbone.conv1.weight' in full and 'encoder_k.1.mlp.2.bias' in full
    np.testing.assert_allclose(full['queue'].norm(dim=0).numpy(), 1.0, atol=1e-5)
    # physical layouts: conv [K][R][S][C], linear [out][in]; grads alias the flat buffer
    w = model.encoder_q[0].layer1[0].conv2.weight
    assert w.shape == (64, 64, 3, 3) and w.stride() == (576, 1, 192, 64)
    lw = model.encoder_q[1].mlp[0].weight
    assert lw.shape == (2048, 2048) and lw.stride() == (1, 2048)
    a = model.arena_q
    assert w.grad.untyped_storage().data_ptr() == a.grads.untyped_storage().data_ptr()
    assert a.n_train == 27966656 and a.total - a.n_train == 2 * 26560
    # key encoder: copy of q, no grads, frozen-statistics BN
    assert torch.equal(model.arena_k.flat, model.arena_q.flat)
    assert all(not p.requires_grad for p in model.encoder_k.parameters())
    assert all(m._use_global_stats for m in model.encoder_k.modules() if hasattr(m, '_use_global_stats'))
    # load_state_dict round trip through the strided views
    sd = {k: torch.randn_like(v) for k, v in qsd.items()}
    model.encoder_q.load_state_dict(sd)
    for k, v in model.encoder_q.state_dict().items():
        assert torch.equal(v, sd[k]), k
    # and the flat buffer holds conv weights physically as KRSC
    off, n = a.param_slices[0]
    assert torch.equal(a.flat[off:off + n].view(64, 7, 7, 3), sd['0.conv1.weight'].permute(0, 2, 3, 1))


def test_cosine_schedule_and_momentum_spelling():
    from passl_amd.solver import build_lr_scheduler, LRSCHEDULERS, OPTIMIZERS
    cfg = AttrDict(name='CosineAnnealingDecay', learning_rate=0.015, T_max=200)
    s = build_lr_scheduler(cfg, iters_per_epoch=10)
    assert s.T_max == 2000 and s.get_lr() == 0.015 and s.last_epoch == 0
    for t in range(1, 5):
        s.step()
        assert abs(s() - 0.015 * 0.5 * (1 + math.cos(math.pi * t / 2000))) < 1e-15
    assert 'Momentum' in OPTIMIZERS and 'LinearWarmup' in LRSCHEDULERS and 'MultiStepDecay' in LRSCHEDULERS
    with pytest.raises(NotImplementedError):
        OPTIMIZERS.get('Momentum')(0.1, parameters=[torch.nn.Parameter(torch.zeros(3))])


def _register_dummies():
    """A tiny CPU model / optimizer registered the way a third-party user would."""
    from passl_amd.modeling.architectures.builder import MODELS
    from passl_amd.solver.builder import OPTIMIZERS
    if 'DummySSL' in MODELS:
        return

    @MODELS.register()
    class DummySSL(torch.nn.Module):
        def __init__(self, dim=4):
            super().__init__()
            self.fc = torch.nn.Linear(3, dim)
            self.calls = []

        def forward(self, xq, xk, mode='train', **kw):
            self.calls.append(sorted(kw))
            return {'loss': (self.fc(xq.mean((2, 3))) - self.fc(xk.mean((2, 3)))).pow(2).mean().reshape(1),
                    'acc1': torch.tensor([50.0])}

    @OPTIMIZERS.register()
    class PlainSGD(object):
        type = 'sgd'

        def __init__(self, learning_rate, parameters=None, **kw):
            self.lr, self.params = learning_rate, list(parameters)

        def clear_grad(self):
            for p in self.params:
                p.grad = None

        def step(self):
            with torch.no_grad():
                for p in self.params:
                    p -= float(self.lr()) * p.grad

        def get_lr(self):
            return float(self.lr())

        def state_dict(self):
            return {'LR_Scheduler': self.lr.state_dict()} if hasattr(self.lr, 'state_dict') else {}

        def set_state_dict(self, sd):
            if 'LR_Scheduler' in sd:
                self.lr.set_state_dict(sd['LR_Scheduler'])


def test_trainer_loop_hooks_and_checkpoint(tmp_path):
    _register_dummies()
    from passl_amd.engine.trainer import Trainer
    from passl_amd.hooks import Hook, HOOKS
    events = []
    if 'SpyHook' not in HOOKS:
        @HOOKS.register()
        class SpyHook(Hook):
            def __init__(self, priority=1):
                self.priority = priority

            def train_iter_begin(self, t):
                events.append(('begin', t.current_iter, t.inner_iter))

            def train_iter_end(self, t):
                events.append(('end', t.current_iter, float(t.lr_scheduler.get_lr())))
        HOOKS.get('SpyHook').events = events
    else:
        events = HOOKS.get('SpyHook').events
        del events[:]
    cfg = get_config(OUR_CFG, ['device=cpu', 'epochs=2', 'dataloader.train.sampler.batch_size=4',
                               'dataloader.train.dataset.num_samples=12',
                               'dataloader.train.dataset.image_size=8',
                               'log_config.interval=1', 'checkpoint.interval=1'])
    cfg.model = AttrDict(name='DummySSL', dim=4)
    cfg.optimizer = AttrDict(name='PlainSGD')
    cfg.custom_config = [AttrDict(name='SpyHook')]
    cfg.output_dir = str(tmp_path)
    cfg.timestamp = '-t'
    tr = Trainer(cfg)
    names = [type(h).__name__ for h in tr.hooks]
    assert names == ['OptimizerHook', 'IterTimerHook', 'CheckpointHook', 'LogHook',
                     'LRSchedulerHook', 'SpyHook']       # trainer.py:235-263 order, stable sort
    assert tr.iters_per_epoch == 3 and tr.total_iters == 6
    w0 = tr.model.fc.weight.detach().clone()
    tr.train()
    assert tr.current_iter == 6 and tr.current_epoch == 2
    assert not torch.equal(w0, tr.model.fc.weight)
    assert [e[:2] for e in events if e[0] == 'begin'] == [('begin', i + 1) for i in range(6)]
    assert [e[2] for e in events if e[0] == 'begin'] == [0, 1, 2, 0, 1, 2]
    # model call contract: total_iters / current_iter / mixup_fn kwargs (trainer.py:326-330)
    assert tr.model.calls[0] == ['current_iter', 'mixup_fn', 'total_iters']
    # cosine lr stepped per iteration: lr seen by the spy at iter i is lr(i) (spy runs after LR hook)
    lrs = [e[2] for e in events if e[0] == 'end']
    assert abs(lrs[0] - 0.015 * 0.5 * (1 + math.cos(math.pi * 1 / 600))) < 1e-12
    # LogHook flushed the deferred device scalars into AverageMeters
    assert os.path.exists(os.path.join(str(tmp_path), 'epoch_2.pd'))
    assert os.path.islink(os.path.join(str(tmp_path), 'latest.pd'))


def test_v2_train_one_step_facade():
    """ContrastiveLearningTrainingEpochLoop.train_one_step(batch) -> (None, loss_dict)
    (passl/engine/loops/contrastive_learning_loop.py:67-88), with gradient accumulation."""
    _register_dummies()
    from types import SimpleNamespace
    from passl_amd.engine.loops import ContrastiveLearningTrainingEpochLoop
    from passl_amd.modeling.architectures.builder import MODELS
    from passl_amd.solver.builder import OPTIMIZERS
    from passl_amd.solver.lr_scheduler import CosineAnnealingDecay
    model = MODELS.get('DummySSL')()
    sched = CosineAnnealingDecay(0.1, T_max=10)
    opt = OPTIMIZERS.get('PlainSGD')(sched, parameters=model.parameters())
    opt._parameter_list = opt.params
    tr = SimpleNamespace(model=model, optimizer=opt, lr_scheduler=sched, accum_steps=2,
                         lr_decay_unit='step')
    loop = ContrastiveLearningTrainingEpochLoop(tr, epochs=1)
    batch = [[torch.randn(4, 3, 8, 8), torch.randn(4, 3, 8, 8)], torch.zeros(4)]
    w0 = model.fc.weight.detach().clone()
    loop.global_step += 1                 # advanced by train_one_epoch (loop.py:282), not by the step itself
    out, loss_dict = loop.train_one_step(batch)
    assert out is None and 'loss' in loss_dict
    assert loop.global_step == 1
    # optimizer.lr_step(self.global_step) (contrastive_learning_loop.py:86-87): the schedule is SET to the step count
    assert not torch.equal(w0, model.fc.weight) and sched.last_epoch == 1
    loop.global_step = 7
    loop.train_one_step(batch)
    assert sched.last_epoch == 7


def test_v2_classification_loops_host_logic(tmp_path):
    """ClassificationTrainingEpochLoop / ClassificationEvaluationLoop (passl/engine/loops/classification_loop.py) on a
    torch-CPU stand-in: micro-batch accumulation equals the full batch, the schedule is set to the step count, the
    train metric joins the logged entries, the evaluation pass weights every batch by its rows (uneven last batch),
    the best metric is tracked, validation after every epoch writes epoch / latest / best checkpoints."""
    _register_dummies()
    import torch.nn.functional as F
    from types import SimpleNamespace
    from passl_amd.engine.loops import ClassificationEvaluationLoop, ClassificationTrainingEpochLoop
    from passl_amd.solver.builder import OPTIMIZERS
    from passl_amd.solver.lr_scheduler import TimmCosine

    class Probe(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.fc = torch.nn.Linear(6, 5)

        def forward(self, x):
            return self.fc(x)

        def save(self, path, local_rank=0, rank=0):
            os.makedirs(os.path.dirname(path), exist_ok=True)
            torch.save(self.state_dict(), path + '.pdparams')

    def loss_func(out, label):
        v = F.cross_entropy(out, label)
        return {'CELoss': v, 'loss': v}

    def metric_func(out, label):
        a = (out.argmax(1) == label).float().mean()
        return {'top1': a, 'metric': a}

    class Loader(list):
        @property
        def dataset(self):
            return range(sum(b[0].shape[0] for b in self))

    gen = torch.Generator().manual_seed(0)
    train = Loader([[torch.randn(8, 6, generator=gen), torch.randint(0, 5, (8,), generator=gen)] for _ in range(3)])
    ev = Loader([[torch.randn(n, 6, generator=gen), torch.randint(0, 5, (n,), generator=gen)] for n in (8, 8, 3)])

    def make(accum):
        torch.manual_seed(1)
        model = Probe()
        sched = TimmCosine(learning_rate=0.5, step_each_epoch=3, epochs=2, decay_unit='step', last_epoch=0)
        opt = OPTIMIZERS.get('PlainSGD')(sched, parameters=model.parameters())
        opt._parameter_list = opt.params
        tr = SimpleNamespace(model=model, optimizer=opt, lr_scheduler=sched, lr_decay_unit='step', accum_steps=accum,
                             train_loss_func=loss_func, eval_loss_func=loss_func, train_metric_func=metric_func,
                             eval_metric_func=metric_func, train_dataloader=train, eval_dataloader=ev,
                             print_batch_step=1, save_interval=5, mode='train', training=True, validating=False,
                             checkpoint=None, output_dir=str(tmp_path / ('accum%d' % accum)), model_name='probe',
                             config={'Global': {'eval_during_train': True, 'eval_interval': 1, 'eval_unit': 'epoch',
                                                'world_size': 1}})
        val = ClassificationEvaluationLoop(tr)
        loop = ClassificationTrainingEpochLoop(tr, epochs=2, val_loop=val)
        tr.cur_epoch_id = 0
        return tr, loop, val

    tr1, loop1, val1 = make(1)
    loop1.global_step = 1
    out, ld = loop1.train_one_step(train[0])
    assert out.shape == (8, 5) and set(ld) == {'CELoss', 'loss', 'top1', 'metric'} and tr1.lr_scheduler.last_epoch == 1
    tr2, loop2, _ = make(2)
    loop2.global_step = 1
    out2, ld2 = loop2.train_one_step(train[0])
    assert torch.allclose(out, out2) and abs(float(ld['loss']) - float(ld2['loss'])) < 1e-6
    assert torch.allclose(tr1.model.fc.weight, tr2.model.fc.weight, atol=1e-6)
    # evaluation: row-weighted averages over 8 + 8 + 3 rows
    tr1.validating = True
    res = val1.run()
    with torch.no_grad():
        xs, ys = torch.cat([b[0] for b in ev]), torch.cat([b[1] for b in ev])
        logits = tr1.model(xs)
        assert abs(res['loss'] - float(F.cross_entropy(logits, ys))) < 1e-6
        assert abs(res['top1'] - float((logits.argmax(1) == ys).float().mean())) < 1e-6
    assert val1.best_model_to_save and val1.best_model_metric == res and tr1.validating is False
    val1.latest_model_metric = dict(res, metric=res['metric'] - 0.1)
    val1.reset_state()
    val1.update_best_model_metric_info()
    assert not val1.best_model_to_save and val1.best_model_metric == res            # a worse pass keeps the best
    # the whole run: validation after each epoch, checkpoints with the metric
    tr, loop, val = make(1)
    loop.run()
    assert loop.global_step == 6 and tr.lr_scheduler.last_epoch == 6 and tr.training is False
    d = os.path.join(tr.output_dir, 'probe')
    for stem in ('epoch_1', 'epoch_2', 'latest', 'best'):
        assert os.path.exists(os.path.join(d, stem + '.pdstates')), os.listdir(d)
    from passl_amd.utils.checkpoint import load_pickle
    st = load_pickle(os.path.join(d, 'latest.pdstates'))
    assert st['epoch'] == 2 and st['global_step'] == 6 and 'top1' in st and 'loss' in st
    # Global.max_num_latest_checkpoint (passl/utils/io.py:172-201): 0 — the task yamls' value — keeps no epoch_*
    # checkpoint (latest / best survive); N keeps the N most recent by their stored timestamp
    import pickle
    for i, ts in ((1, '2024-01-01 00:00:01'), (2, '2024-01-01 00:00:02')):
        meta = load_pickle(os.path.join(d, 'epoch_%d.pdstates' % i))
        meta['timestamp'] = ts
        with open(os.path.join(d, 'epoch_%d.pdstates' % i), 'wb') as f:
            pickle.dump(meta, f, protocol=2)
    # resume (loop.py:358-375): counters and the stored metric — the bar a later evaluation must pass to become 'best'
    tr3, loop3, val3 = make(1)
    tr3.checkpoint = os.path.join(d, 'latest')
    tr3.model.load_pretrained = lambda path, rank=0, finetune=False: tr3.model.load_state_dict(torch.load(path + '.pdparams'))
    loop3.resume()
    assert loop3.start_eopch == 2 and loop3.global_step == 6 and val3.best_model_metric['metric'] == st['metric']
    tr.config['Global']['max_num_latest_checkpoint'] = 1
    loop._prune_checkpoints(d)
    left = sorted(os.listdir(d))
    assert 'epoch_2.pdstates' in left and 'epoch_2.pdparams' in left and not any(n.startswith('epoch_1.') for n in left)
    tr.config['Global']['max_num_latest_checkpoint'] = 0
    loop._prune_checkpoints(d)
    left = sorted(os.listdir(d))
    assert not any(n.startswith('epoch_') for n in left) and 'latest.pdparams' in left and 'best.pdstates' in left


# ------------------------------------------------------------------ SimCLR row (host side)
REF_SIMCLR_CFG = '/root/reference/configs/simclr/simclr_r50_IM.yaml'


@pytest.mark.skipif(not os.path.exists(REF_SIMCLR_CFG), reason='reference tree not present')
def test_reference_simclr_config_loads_and_builds_unchanged():
    hip_config.set_device('cpu')
    from passl_amd.modeling import build_model
    from passl_amd.solver import build_lr_scheduler_simclr, build_optimizer
    from oracle.simclr import init_encoder_state
    cfg = get_config(REF_SIMCLR_CFG, [])
    assert cfg.use_simclr_iters and cfg.global_batch_size == 4096
    model = build_model(cfg.model)
    assert model.head.temperature == 0.1 and model.head.multi_rank is False
    assert not hasattr(model.backbone, 'maxpool') and model.backbone.with_pool
    # state_dict layout = the oracle's (= the reference's own, checked in oracle/ref_runner)
    ost = init_encoder_state(torch.Generator().manual_seed(0))
    sd = model.encoder.state_dict()
    assert list(sd.keys()) == list(ost.keys())
    assert all(tuple(sd[k].shape) == tuple(ost[k].shape) for k in ost)
    assert sum(p.numel() for p in model.parameters() if p.requires_grad) == 32171456
    # trainer.py:157-163 passes batch_size*8 ; builder.py:54-66
    sched = build_lr_scheduler_simclr(cfg.lr_scheduler, 2502, 512 * 8, cfg.epochs, 0)
    assert (sched.warmup_steps, sched.learning_rate.T_max, sched.end_lr) == (3127, 28152, 64.0)
    assert sched() == 0.0
    for _ in range(3127):
        sched.step()
    assert abs(sched() - 64.0) < 1e-9
    sched.step()
    assert abs(sched() - 64.0 * (1 + math.cos(math.pi * 1 / 28152)) / 2) < 1e-9
    opt = build_optimizer(cfg.optimizer, sched, [model])
    assert 'lars' in opt.type and opt._wd == 1e-4 and opt._coeff == 0.001 and opt._momentum == 0.9
    # exclude_from_weight_decay is matched against Paddle auto-names: nothing matches
    assert opt.param_names[0] == 'conv2d_0.w_0' and opt.param_names[1] == 'batch_norm2d_0.w_0'
    assert float(opt._tables[0]['seg_wd'].min()) > 0
    t = opt._tables[0]
    assert int(t['blk_len'].sum()) == sum(n for _o, n in model.arena_q.param_slices)
    assert int(t['blk_len'].max()) <= 4096 and int((t['blk_off'] % 4).sum()) == 0


def test_simclr_registry_names_and_kwargs():
    from passl_amd.modeling.architectures import MODELS
    from passl_amd.modeling.backbones import BACKBONES
    from passl_amd.modeling.heads import HEADS
    from passl_amd.modeling.necks import NECKS
    from passl_amd.solver import LRSCHEDULERS, OPTIMIZERS
    for reg, names in ((MODELS, ['MoCo', 'SimCLR']), (BACKBONES, ['ResNet', 'ResNetsimclr']),
                       (NECKS, ['LinearNeck', 'NonLinearNeckV1', 'NonLinearNeckfc3']),
                       (HEADS, ['ContrastiveHead', 'SimCLRContrastiveHead']),
                       (OPTIMIZERS, ['Momentum', 'LarsMomentumOptimizer']),
                       (LRSCHEDULERS, ['CosineAnnealingDecay', 'simclrCosineWarmup', 'Cosinesimclr'])):
        for n in names:
            assert n in reg, n
    h = HEADS.get('SimCLRContrastiveHead')(temperature=0.5, return_accuracy=True, multi_rank=False)
    assert h.temperature == 0.5 and h.co2_weight == 3.0
    with pytest.raises(NotImplementedError):
        hip_config.set_device('cpu')
        BACKBONES.get('ResNetsimclr')(depth=20)


# ------------------------------------------------------------------ checkpoint / weight exchange
def test_checkpoint_file_format_and_extract_weight(tmp_path):
    """CheckpointHook writes the reference's pickle-of-numpy layout (hooks/checkpoint_hook.py:23-50);
    tools/extract_weight.py applies the reference's prefix rules (tools_v110/extract_weight.py)."""
    import pickle
    import sys
    from types import SimpleNamespace
    hip_config.set_device('cpu')
    import moco_util as U
    from passl_amd.hooks.checkpoint_hook import CheckpointHook
    from passl_amd.modeling import build_model
    from passl_amd.solver.lr_scheduler import CosineAnnealingDecay
    from passl_amd.solver.optimizer import Momentum
    from passl_amd.utils.checkpoint import load_pickle, load_state_into
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import extract_weight as EW
    cfg = dict(U.MODEL_CFG)
    cfg.update(K=256)
    torch.manual_seed(3)
    model = build_model(cfg)
    sched = CosineAnnealingDecay(0.015, T_max=100)
    opt = Momentum(sched, parameters=list(model.parameters()), weight_decay=1e-4)
    tr = SimpleNamespace(model=model, optimizer=opt, lr_scheduler=sched, current_epoch=0, rank=0,
                         output_dir=str(tmp_path), logger=SimpleNamespace(info=lambda *a: None))
    hook = CheckpointHook(interval=1)
    hook.every_n_epochs = lambda t, n: True
    for ep in range(7):                       # max_keep_ckpts = 5
        tr.current_epoch = ep
        hook.train_epoch_end(tr)
    files = sorted(os.listdir(tmp_path))
    assert files == ['epoch_3.pd', 'epoch_4.pd', 'epoch_5.pd', 'epoch_6.pd', 'epoch_7.pd', 'latest.pd']
    assert os.readlink(os.path.join(tmp_path, 'latest.pd')) == 'epoch_7.pd'
    with open(os.path.join(tmp_path, 'latest.pd'), 'rb') as f:
        ck = pickle.load(f)
    assert set(ck) == {'epoch', 'state_dict', 'optimizer', 'lr_scheduler'} and ck['epoch'] == 7
    sd = ck['state_dict']
    assert all(isinstance(v, np.ndarray) for v in sd.values())
    assert sd['encoder_q.0.conv1.weight'].shape == (64, 3, 7, 7)
    assert sd['encoder_q.1.mlp.0.weight'].shape == (2048, 2048)            # Paddle Linear [in, out]
    assert sd['encoder_q.0.bn1._variance'].shape == (64,) and sd['queue'].shape == (128, 256)
    assert sd['queue_ptr'].dtype == np.int64
    np.testing.assert_array_equal(sd['backbone.conv1.weight'], sd['encoder_q.0.conv1.weight'])  # moco.py:65 alias
    # extract_weight: prefix filter, --remove_prefix, error when the prefix is absent
    out = EW.extract(ck, 'backbone', remove_prefix=True)
    assert 'conv1.weight' in out and 'layer4.2.bn3._mean' in out and not any(k.startswith('backbone') for k in out)
    assert set(EW.extract(ck, 'encoder_k')) == {k for k in sd if k.startswith('encoder_k')}
    with pytest.raises(Exception, match='Cannot find a nothere layer'):
        EW.extract(ck, 'nothere')
    EW.main([os.path.join(tmp_path, 'latest.pd'), '--prefix', 'backbone', '--remove_prefix', '--output',
             os.path.join(tmp_path, 'bb.pdparams')])
    # a paddle.save()-style file (bookkeeping key) loads into a fresh backbone without transposes
    with open(os.path.join(tmp_path, 'bb.pdparams'), 'rb') as f:
        bb = pickle.load(f)
    bb['StructuredToParameterName@@'] = {'conv1.weight': 'conv2d_0.w_0'}
    with open(os.path.join(tmp_path, 'bb2.pdparams'), 'wb') as f:
        pickle.dump(bb, f, protocol=2)
    torch.manual_seed(9)
    model2 = build_model(cfg)
    missing, unexpected = load_state_into(model2.backbone, load_pickle(os.path.join(tmp_path, 'bb2.pdparams')))
    assert not missing and not unexpected
    for k, v in model.backbone.state_dict().items():
        assert torch.equal(model2.backbone.state_dict()[k], v), k
    # optimizer / scheduler round trip
    opt._velocity[0].normal_()
    for _ in range(5):
        sched.step()
    sched2 = CosineAnnealingDecay(0.015, T_max=100)
    opt2 = Momentum(sched2, parameters=list(model2.parameters()), weight_decay=1e-4)
    from passl_amd.utils.checkpoint import to_numpy, to_tensors
    opt2.set_state_dict(to_tensors(to_numpy(opt.state_dict())))
    assert torch.equal(opt2._velocity[0], opt._velocity[0]) and sched2.last_epoch == 5 and sched2() == sched()


# ------------------------------------------------------------------ MAE / CLIP rows (host side)
REF_MAE_CFG = '/root/reference/configs/mae/mae_vit_b_pretrain.yaml'
REF_CLIP_CFG = '/root/reference/configs/clip/vit-b-32.yaml'


@pytest.mark.skipif(not os.path.exists(REF_MAE_CFG), reason='reference tree not present')
def test_reference_mae_config_loads_and_builds_unchanged():
    hip_config.set_device('cpu')
    from passl_amd.modeling import build_model
    from passl_amd.solver import build_lr_scheduler, build_optimizer
    from oracle.mae import init_state, VIT_B
    cfg = get_config(REF_MAE_CFG, [])
    model = build_model(cfg.model)
    arch = {k: VIT_B[k] for k in ('img_size', 'patch_size', 'embed_dim', 'depth', 'decoder_embed_dim',
                                  'decoder_depth', 'mlp_ratio')}
    ost = init_state(torch.Generator().manual_seed(0), **arch)
    sd = model.backbone.state_dict()
    assert set(sd.keys()) == set(ost.keys())
    assert all(tuple(sd[k].shape) == tuple(ost[k].shape) for k in ost)
    assert sum(p.numel() for p in model.parameters() if p.requires_grad) == 111655680
    sched = build_lr_scheduler(cfg.lr_scheduler, 10)
    opt = build_optimizer(cfg.optimizer, sched, [model])
    assert opt.type == 'adamw' and (opt._b1, opt._b2, opt._wd) == (0.9, 0.95, 0.05)


@pytest.mark.skipif(not os.path.exists(REF_CLIP_CFG), reason='reference tree not present')
def test_reference_clip_config_loads_and_builds_unchanged():
    hip_config.set_device('cpu')
    from passl_amd.modeling import build_model
    from passl_amd.solver import build_lr_scheduler, build_optimizer
    from oracle.clip import init_state, VIT_B_32
    cfg = get_config(REF_CLIP_CFG, [])
    model = build_model(cfg.model)
    ost = init_state(torch.Generator().manual_seed(0), VIT_B_32)
    sd = model.model.state_dict()
    # same keys as the reference's CLIP.state_dict() (pinned in oracle/ref_runner.load_clip_state),
    # including the raw matrices `visual.proj` / `text_projection` that run as bias-free Linears here
    assert set(sd.keys()) == set(ost.keys())
    assert all(tuple(sd[k].shape) == tuple(ost[k].shape) for k in ost)
    assert sum(p.numel() for p in model.parameters()) == sum(v.numel() for v in ost.values()) == 151277313
    assert model.model.transformer.blocks[0].attn.causal and not model.model.visual.blocks[0].attn.causal
    assert abs(float(model.model.logit_scale.detach()) - math.log(1 / 0.07)) < 1e-6
    # reference init (clip.py:271-282): text proj std = width^-0.5 * (2 * depth)
    w = model.model.transformer.blocks[3].attn.proj.weight
    assert abs(float(w.std()) - 512 ** -0.5 * 24) < 0.02
    sched = build_lr_scheduler(cfg.lr_scheduler, 591)
    opt = build_optimizer(cfg.optimizer, sched, [model])
    assert opt.type == 'adamw' and (opt._b1, opt._b2, opt._eps, opt._wd) == (0.9, 0.98, 1e-8, 0.0005)
    # state_dict round trip through the alias hooks
    sd2 = {k: v.clone() for k, v in model.state_dict().items()}
    assert 'model.visual.proj' in sd2 and 'model.visual.proj.weight' not in sd2


def test_clip_mask_and_registry_guards():
    hip_config.set_device('cpu')
    from passl_amd.modeling.backbones import BACKBONES
    from passl_amd.modeling.backbones.vision_transformer import _is_causal
    from passl_amd.modeling.heads import HEADS
    assert 'CLIP' in BACKBONES and 'VisionTransformer' in BACKBONES and 'CLIPHead' in HEADS
    T = 5
    causal = torch.triu(torch.full((T, T), -math.inf), 1)
    assert _is_causal(causal) and _is_causal('causal') and not _is_causal(None)
    with pytest.raises(NotImplementedError):
        _is_causal(torch.zeros(T, T))
    with pytest.raises(NotImplementedError):          # ModifiedResNet tower is not built
        BACKBONES.get('CLIP')(embed_dim=64, image_resolution=64, vision_layers=(1, 1, 1, 1), vision_width=64,
                              vision_patch_size=None, pre_norm=True, proj=True, patch_bias=False,
                              context_length=8, vocab_size=50, transformer_width=64, transformer_heads=1,
                              transformer_layers=1, qkv_bias=True)
    head = HEADS.get('CLIPHead')()
    a = torch.zeros(4, 4)
    from passl_amd.hip.lib import PasslHipError
    with pytest.raises(PasslHipError):                # two separate matrices = the cross-rank form: a HIP
        head(a, torch.zeros(4, 4), torch.arange(4), torch.arange(4))      # path, refused on host tensors
    with pytest.raises(ValueError):                   # one label per row
        head(a, torch.zeros(4, 4), torch.arange(3), torch.arange(4))
    with pytest.raises(NotImplementedError):          # the fused (logits, logits.t()) kernel has arange(B) built in
        head(a, a.t(), torch.tensor([0, 1, 3, 2]), torch.arange(4))


def test_synthetic_image_text_dataset():
    from passl_amd.datasets.builder import build_dataloader
    cfg = dict(dataset=dict(name='SyntheticImageText', num_samples=64, image_size=32, context_length=12,
                            vocab_size=100), sampler=dict(batch_size=8))
    loader, _ = build_dataloader(cfg, 'cpu')
    image, text = next(iter(loader))
    assert image.shape == (8, 3, 32, 32) and text.shape == (8, 12) and text.dtype == torch.int64
    am = text.argmax(dim=-1)
    assert bool((text[torch.arange(8), am] == 99).all()) and int(text.max()) == 99 and int(text.min()) == 0
    assert all(int(text[b, am[b] + 1:].abs().sum()) == 0 for b in range(8))
    assert len(loader) == 8


# ------------------------------------------------------------------ linear-probe row (host side)
REF_CLAS_CFG = '/root/reference/configs/moco/moco_clas_r50.yaml'


@pytest.mark.skipif(not os.path.exists(REF_CLAS_CFG), reason='reference tree not present')
def test_reference_linear_probe_config_loads_and_builds_unchanged():
    hip_config.set_device('cpu')
    from passl_amd.hooks import build_hook, EvaluateHook
    from passl_amd.modeling import build_model
    from passl_amd.solver import build_lr_scheduler, build_optimizer
    from oracle.clas import init_state
    cfg = get_config(REF_CLAS_CFG, [])
    model = build_model(cfg.model)
    ost = init_state(torch.Generator().manual_seed(0))
    sd = model.state_dict()
    assert set(sd.keys()) == set(ost.keys())
    assert all(tuple(sd[k].shape) == tuple(ost[k].shape) for k in ost)
    # frozen_stages = 4: only the fc is trainable, every BatchNorm uses its running statistics
    assert sorted(n for n, p in model.named_parameters() if p.requires_grad) == ['head.fc_cls.bias',
                                                                                 'head.fc_cls.weight']
    assert model.backbone.fully_frozen and model.backbone._all_bn_frozen()
    sched = build_lr_scheduler(cfg.lr_scheduler, 100)          # milestones are epochs -> iterations
    assert sched() == 30.0
    for _ in range(6000):
        sched.step()
    assert abs(sched() - 3.0) < 1e-9
    opt = build_optimizer(cfg.optimizer, sched, [model])
    assert opt.type == 'momentum' and opt._wd == 0.0 and opt._momentum == 0.9 and len(opt._parameter_list) == 2
    assert isinstance(build_hook(dict(cfg.custom_config[0])), EvaluateHook)
    # partially frozen trunk (resnet.py:90-106): stem + layer1-2 frozen, the rest trains with the head
    m2 = build_model(dict(name='Classification', backbone=dict(name='ResNet', depth=50, frozen_stages=2),
                          head=dict(name='ClasHead', with_avg_pool=True, in_channels=2048)))
    tr = {n for n, p in m2.named_parameters() if p.requires_grad}
    assert 'backbone.layer3.0.conv1.weight' in tr and 'backbone.layer4.2.bn3.bias' in tr and 'head.fc_cls.weight' in tr
    assert not any(n.startswith(('backbone.conv1', 'backbone.bn1', 'backbone.layer1', 'backbone.layer2')) for n in tr)
    bb = m2.backbone
    assert bb.layer2[0].bn1.uses_global_stats() and not bb.layer3[0].bn1.uses_global_stats() and not bb.fully_frozen
    assert m2.arena_k is not None and not m2.arena_k.trainable and m2.arena_q.trainable
    assert sum(n for _o, n in m2.arena_q.param_slices) == sum(p.numel() for p in m2.parameters() if p.requires_grad)
    assert sorted(m2.state_dict()) == sorted(model.state_dict())           # same checkpoint keys
    # a trainable trunk (supervised training) = one arena for trunk + head, no frozen arena
    m3 = build_model(dict(name='Classification', backbone=dict(name='ResNet', depth=50),
                          head=dict(name='ClasHead', with_avg_pool=True, in_channels=2048)))
    assert m3.arena_k is None and all(p.requires_grad for p in m3.parameters())
    # the pre-training architectures keep ONE trainable arena: a frozen prefix there is refused loudly
    with pytest.raises(NotImplementedError):
        build_model(dict(name='MoCo', backbone=dict(name='ResNet', depth=50, frozen_stages=1),
                         neck=dict(name='NonLinearNeckV1', in_channels=2048, hid_channels=2048, out_channels=128,
                                   with_avg_pool=True), head=dict(name='ContrastiveHead', temperature=0.2)))


def test_synthetic_labeled_dataset_and_evaluate_contract():
    from passl_amd.datasets.builder import build_dataloader
    cfg = dict(dataset=dict(name='SyntheticLabeled', num_samples=64, image_size=32, num_classes=10),
               sampler=dict(batch_size=8, drop_last=False))
    loader, _ = build_dataloader(cfg, 'cpu')
    image, label = next(iter(loader))
    assert image.shape == (8, 3, 32, 32) and label.shape == (8,) and label.dtype == torch.int64
    assert int(label.min()) >= 0 and int(label.max()) < 10 and len(loader) == 8 and len(loader.dataset) == 64
    assert hasattr(loader.dataset, 'evaluate')


REF_MOCO_V1_CFG = '/root/reference/configs/moco/moco_v1_r50.yaml'


@pytest.mark.skipif(not os.path.exists(REF_MOCO_V1_CFG), reason='reference tree not present')
def test_reference_moco_v1_config_loads_and_builds_unchanged():
    """configs/moco/moco_v1_r50.yaml: LinearNeck projector, T = 0.07, MultiStepDecay(0.03, [120, 160])."""
    hip_config.set_device('cpu')
    from passl_amd.modeling import build_model
    from passl_amd.solver import build_lr_scheduler, build_optimizer
    from oracle.resnet50 import init_encoder_state
    cfg = get_config(REF_MOCO_V1_CFG, [])
    model = build_model(cfg.model)
    assert type(model.encoder_q[1]).__name__ == 'LinearNeck' and model.head.temperature == 0.07
    ost = init_encoder_state(torch.Generator().manual_seed(0), neck='LinearNeck')
    sd = model.encoder_q.state_dict()
    assert list(sd.keys()) == list(ost.keys())
    assert all(tuple(sd[k].shape) == tuple(ost[k].shape) for k in ost)
    sched = build_lr_scheduler(cfg.lr_scheduler, 5004)
    assert sched() == 0.03
    for _ in range(120 * 5004):
        sched.last_epoch += 1
    assert abs(sched.get_lr() - 0.003) < 1e-12
    opt = build_optimizer(cfg.optimizer, sched, [model])
    assert opt.type == 'momentum' and opt._wd == 1e-4


# ------------------------------------------------------------------ `passl` import name + v2 façade
def test_passl_alias_resolves_reference_imports():
    """tools_v110/train.py:22-25 style imports bind, and they are the SAME module objects as passl_amd's
    (one set of registries)."""
    from passl.utils.options import parse_args          # noqa: F401
    from passl.utils.config import get_config as gc2
    from passl.utils.setup import setup                  # noqa: F401
    from passl.engine.trainer import Trainer as T2
    import passl.modeling.architectures.builder as b1
    import passl_amd.modeling.architectures.builder as b2
    from passl_amd.engine.trainer import Trainer as T1
    assert T1 is T2 and gc2 is get_config and b1.MODELS is b2.MODELS
    # v2 surface + the loss homes named by BASELINE.json north_star
    from passl.models import build_model, Model          # noqa: F401
    from passl.engine.engine import Engine               # noqa: F401
    from passl.engine.loops import ContrastiveLearningTrainingEpochLoop, TrainingEpochLoop
    assert issubclass(ContrastiveLearningTrainingEpochLoop, TrainingEpochLoop)
    from passl.core import grad_sync, param_sync         # noqa: F401
    import passl.loss.moco, passl.loss.nt_xent, passl.loss.mae
    assert callable(passl.loss.moco.info_nce) and callable(passl.loss.nt_xent.nt_xent)
    assert callable(passl.loss.mae.masked_patch_loss)
    with pytest.raises(ImportError):
        import passl.no_such_module                       # noqa: F401
    # the heads call into the loss homes
    from passl_amd.modeling.heads import simclr_contrastive_head as h
    assert h._NTXentFn is passl.loss.nt_xent._NTXentFn


def test_v2_engine_runs_the_named_loop(tmp_path):
    """Engine(config).train(): v2 schema -> model via passl.models.build_model -> loop resolved by name
    -> run / train_one_epoch / train_one_step with accumulation, lr stepping and max_train_step."""
    _register_dummies()
    import passl_amd.models as M
    from passl_amd.engine.engine import Engine
    from passl_amd.modeling.architectures.builder import MODELS

    class DummyV2(M.Model):
        def __init__(self, dim=4):
            super().__init__()
            self.inner = MODELS.get('DummySSL')(dim)
            self.saved = []

        def forward(self, inputs):
            return self.inner(*inputs)

        def save(self, path, local_rank=0, rank=0):
            self.saved.append(path)
    M.dummy_v2 = lambda **kw: DummyV2(**kw)
    try:
        cfg = get_config(os.path.join(ROOT, 'configs/v2/moco_v2_resnet50_pt_synthetic.yaml'),
                         ['Global.device=cpu', 'Global.epochs=2', 'Global.accum_steps=2', 'Global.save_interval=1',
                          'Global.print_batch_step=1', 'Global.output_dir=%s' % tmp_path,
                          'DataLoader.Train.sampler.batch_size=4', 'DataLoader.Train.dataset.num_samples=12',
                          'DataLoader.Train.dataset.image_size=8'])
        cfg.Model = AttrDict(name='dummy_v2', dim=4)
        cfg.Optimizer = AttrDict(name='PlainSGD')
        eng = Engine(cfg, mode='train')
        eng.optimizer._parameter_list = eng.optimizer.params
        assert eng.accum_steps == 2 and eng.lr_decay_unit == 'step' and eng.grad_reducer is None
        assert type(eng.train_loop).__name__ == 'ContrastiveLearningTrainingEpochLoop'
        assert eng.lr_scheduler.T_max == 2 * 3            # decays over epochs x iterations
        w0 = eng.model.inner.fc.weight.detach().clone()
        eng.train()
        assert eng.global_step == 6 and eng.cur_epoch_id == 2 and eng.lr_scheduler.last_epoch == 6
        assert not torch.equal(w0, eng.model.inner.fc.weight)
        # save_interval 1 -> epoch_N and `latest` every epoch
        assert [os.path.basename(q) for q in eng.model.saved] == ['epoch_1', 'latest', 'epoch_2', 'latest']
        # each optimizer step saw accum_steps micro-batches of 2 samples
        assert len(eng.model.inner.calls) == 12
        # max_train_step ends the run early
        cfg.Global.max_train_step = 4
        eng2 = Engine(cfg, mode='train')
        eng2.optimizer._parameter_list = eng2.optimizer.params
        eng2.train()
        assert eng2.global_step == 4
        with pytest.raises(AttributeError):
            M.build_model(dict(name='no_such_model'))
    finally:
        del M.dummy_v2


def test_v2_checkpoint_set_and_resume(tmp_path):
    """save_checkpoint writes epoch_N.{pdparams (Model.save), pdopt, pdstates} and `latest.*` (reference
    passl/utils/io.py:115-170); Global.checkpoint resumes model, optimizer / lr-scheduler state and the epoch /
    step counters (loop.py:358-375): a run stopped after epoch 1 and resumed ends where the uninterrupted run ends."""
    import pickle
    _register_dummies()
    import passl_amd.models as M
    from passl_amd.engine.engine import Engine
    from passl_amd.modeling.architectures.builder import MODELS
    from passl_amd.solver.lr_scheduler import LRScheduler

    class DummyV2(M.Model):
        def __init__(self, dim=4):
            super().__init__()
            self.inner = MODELS.get('DummySSL')(dim)

        def forward(self, inputs):
            return self.inner(*inputs)

        def save(self, path, local_rank=0, rank=0):
            os.makedirs(os.path.dirname(path), exist_ok=True)
            with open(path + '.pdparams', 'wb') as f:
                pickle.dump({k: v.detach().numpy() for k, v in self.state_dict().items()}, f)

        def load_pretrained(self, path, rank=0, finetune=False):
            with open(path + '.pdparams', 'rb') as f:
                self.load_state_dict({k: torch.as_tensor(v) for k, v in pickle.load(f).items()})
    M.dummy_v2 = lambda **kw: DummyV2(**kw)
    try:
        def make(out, epochs, checkpoint=None):
            cfg = get_config(os.path.join(ROOT, 'configs/v2/moco_v2_resnet50_pt_synthetic.yaml'),
                             ['Global.device=cpu', 'Global.epochs=%d' % epochs, 'Global.save_interval=1',
                              'Global.print_batch_step=100', 'Global.output_dir=%s' % out, 'Global.seed=3',
                              'DataLoader.Train.sampler.batch_size=4', 'DataLoader.Train.dataset.num_samples=12',
                              'DataLoader.Train.dataset.image_size=8'])
            cfg.Model = AttrDict(name='dummy_v2', dim=4)
            cfg.Optimizer = AttrDict(name='PlainSGD')
            cfg.LRScheduler.T_max = 6                      # the schedule of the WHOLE (2-epoch) run in every arm
            cfg.Global.checkpoint = checkpoint
            eng = Engine(cfg, mode='train')
            eng.optimizer._parameter_list = eng.optimizer.params
            return eng
        full = make(tmp_path / 'full', 2)
        full.train()
        first = make(tmp_path / 'part', 1)
        first.train()
        d = tmp_path / 'part' / 'dummy_v2'
        for stem in ('epoch_1', 'latest'):
            for ext in ('.pdparams', '.pdopt', '.pdstates'):
                assert (d / (stem + ext)).exists(), stem + ext
        with open(d / 'latest.pdstates', 'rb') as f:
            st = pickle.load(f)
        assert st['epoch'] == 1 and st['global_step'] == 3 and 'timestamp' in st
        with open(d / 'latest.pdopt', 'rb') as f:
            assert pickle.load(f)['LR_Scheduler']['last_epoch'] == 3
        second = make(tmp_path / 'part', 2, checkpoint=str(d / 'latest'))
        second.train()
        assert second.train_loop.start_eopch == 1 and second.global_step == 6 and second.cur_epoch_id == 2
        assert second.lr_scheduler.last_epoch == full.lr_scheduler.last_epoch == 6
        assert isinstance(second.lr_scheduler, LRScheduler)
        assert torch.equal(second.model.inner.fc.weight, full.model.inner.fc.weight)
    finally:
        del M.dummy_v2


def test_timm_cosine_schedule_known_answers():
    """passl/scheduler/lr_scheduler.py:22-77 with the MoCo-v3 yaml's settings: no step() in the constructor, and the
    value an optimizer sees is get_lr() at the current last_epoch (passl/optimizer/optimizer.py:117-120) — the first
    optimizer step runs at get_lr(-1) = warmup_start_lr; linear warm-up over warmup_epoch epochs of steps, then a
    cosine over the REMAINING steps (warmup_prefix).  Pinned by the reference's own class:
    tests/test_oracle_linprobe_v2.py + tests/golden/make_golden_linprobe_v2.py."""
    from passl_amd.solver.lr_scheduler import TimmCosine
    s = TimmCosine(learning_rate=0.0024, step_each_epoch=10, epochs=30, decay_unit='step', eta_min=0.0,
                   warmup_epoch=4, warmup_start_lr=0.0, warmup_prefix=True)
    assert s.T_max == 300 and s.warmup_steps == 40 and s.last_epoch == -1
    assert s() == 0.0 and s.last_lr == 0.0024             # get_lr(-1) = warmup_start_lr; the cached value is not read
    s.step()
    assert s.last_epoch == 0 and s() == 0.0
    for _ in range(10):
        s.step()
    assert abs(s() - 0.0024 * 10 / 40) < 1e-15
    for _ in range(30):
        s.step()
    assert s.last_epoch == 40 and abs(s() - 0.0024) < 1e-15              # end of the warm-up = start of the cosine
    for _ in range(130):
        s.step()
    assert abs(s() - 0.5 * 0.0024 * (1 + math.cos(math.pi * 130 / 260))) < 1e-15
    for _ in range(130):
        s.step()
    assert s.last_epoch == 300 and abs(s()) < 1e-15
    # epoch unit, no prefix: the cosine is counted from 0 over T_max
    e = TimmCosine(learning_rate=1.0, step_each_epoch=10, epochs=20, decay_unit='epoch', warmup_epoch=5)
    e.step(10)
    assert abs(e() - 0.5 * (1 + math.cos(math.pi * 10 / 20))) < 1e-15
    sd = s.state_dict()
    t = TimmCosine(learning_rate=0.0024, step_each_epoch=10, epochs=30, decay_unit='step', warmup_epoch=4)
    t.set_state_dict(sd)
    assert t.last_epoch == 300


@pytest.mark.skipif(not os.path.isdir('/root/reference/tasks/ssl/mocov3'), reason='reference tree not present')
def test_v2_engine_builds_from_the_reference_mocov3_yaml_unchanged():
    """tasks/ssl/mocov3/configs/mocov3_vit_base_patch16_224_pt_in1k_4n32c_dp_fp16o1.yaml read as it is (only `-o`
    overrides: device, the dataset class — there is no ImageNet here — and a smaller run): Engine builds the ViT-B
    MoCo-v3 model through passl.models.build_model, AdamW from `betas / eps / use_master_param /
    exp_avg_force_fp32`, TimmCosine from the run length, the FP16 section selects bf16 compute, and
    runtime_info_hub carries max_steps to the momentum schedule."""
    from passl_amd.engine.engine import Engine
    from passl_amd.hip import config as hip_config
    from passl_amd.utils.infohub import runtime_info_hub
    yaml_path = '/root/reference/tasks/ssl/mocov3/configs/mocov3_vit_base_patch16_224_pt_in1k_4n32c_dp_fp16o1.yaml'
    prev = hip_config.get_compute_dtype()
    try:
        cfg = get_config(yaml_path, ['Global.device=cpu', 'Global.epochs=50',
                                     'DataLoader.Train.dataset.name=SyntheticTwoView',
                                     'DataLoader.Train.sampler.batch_size=2'])
        cfg.DataLoader.Train.dataset.num_samples = 40          # (not a key of the yaml: `-o` cannot add it)
        cfg.DataLoader.Train.dataset.image_size = 224
        eng = Engine(cfg, mode='train')
        assert hip_config.get_compute_dtype() == torch.bfloat16
        m = eng.model
        assert type(m).__name__ == 'MoCoV3Pretrain' and m.T == 0.2 and m.momentum_encoder.momentum == 0.99
        n_train = sum(p.numel() for p in m.parameters() if p.requires_grad)
        # ViT-B without its (frozen) patch embedding and fixed position table, + projector + predictor
        vit = 768 + 12 * (2 * 768 * 2 + 768 * 2304 + 2304 + 768 * 768 + 768 + 2 * 768 * 3072 + 3072 + 768) + 2 * 768
        proj = 768 * 4096 + 2 * 4096 + 4096 * 4096 + 2 * 4096 + 4096 * 256
        pred = 256 * 4096 + 2 * 4096 + 4096 * 256
        assert n_train == vit + proj + pred == m.arena_q.n_train - sum(
            (-n) % 8 for _o, n in m.arena_q.param_slices)
        opt = eng.optimizer
        assert type(opt).__name__ == 'AdamW' and (opt._b1, opt._b2, opt._eps, opt._wd) == (0.9, 0.999, 1e-8, 0.1)
        assert opt._arenas == [m.arena_q]
        sch = eng.lr_scheduler
        assert type(sch).__name__ == 'TimmCosine' and sch.T_max == 50 * 20 and sch.warmup_steps == 40 * 20
        assert sch.warmup_prefix is True and eng.lr_decay_unit == 'step' and opt.get_lr() == 0.0 \
            and sch.base_lr == 0.0024
        assert runtime_info_hub.max_steps == 1000 and runtime_info_hub.epochs == 50
        assert abs(m.momentum_encoder.current_momentum(500) - 0.99 * 0.5) < 1e-12
        assert type(eng.train_loop).__name__ == 'ContrastiveLearningTrainingEpochLoop'
        sd = m.state_dict()
        assert 'momentum_encoder.model.0.blocks.11.mlp.fc2.weight' in sd and 'predictor.4._variance' in sd
        assert 'base_encoder.head.7.weight' not in sd and tuple(sd['base_encoder.pos_embed'].shape) == (1, 197, 768)
    finally:
        hip_config.set_compute_dtype(prev)


@pytest.mark.skipif(not os.path.isdir('/root/reference/tasks/ssl/simsiam'), reason='reference tree not present')
def test_v2_engine_builds_from_the_reference_simsiam_yaml_unchanged():
    """tasks/ssl/simsiam/configs/simsiam_resnet50_pt_in1k_1n8c_dp_fp32.yaml read as it is (`-o` overrides: device,
    dataset class, a smaller run): Engine builds SimSiam through passl.models.build_model; the schedule sits INSIDE
    the Optimizer block (`lr:` = TimmCosine, stepped per epoch: lr_decay_unit) and `param_groups` split the
    trainable parameters by name into the encoder group (schedule) and the predictor group (fixed 0.1) — one flat
    optimizer per group; FP16 level O0 selects fp32 compute."""
    from passl_amd.engine.engine import Engine, OptimizerGroup
    from passl_amd.hip import config as hip_config
    yaml_path = '/root/reference/tasks/ssl/simsiam/configs/simsiam_resnet50_pt_in1k_1n8c_dp_fp32.yaml'
    prev = hip_config.get_compute_dtype()
    try:
        cfg = get_config(yaml_path, ['Global.device=cpu', 'Global.epochs=10',
                                     'DataLoader.Train.dataset.name=SyntheticTwoView',
                                     'DataLoader.Train.sampler.batch_size=2'])
        cfg.DataLoader.Train.dataset.num_samples = 8
        cfg.DataLoader.Train.dataset.image_size = 32
        eng = Engine(cfg, mode='train')
        assert hip_config.get_compute_dtype() == torch.float32
        m, opt = eng.model, eng.optimizer
        assert type(m).__name__ == 'SimSiamPretain' and isinstance(opt, OptimizerGroup)
        assert opt.names == ['encoder', 'predictor']
        enc, pred = opt.optimizers
        assert enc._arenas == [m.arena_q] and pred._arenas == [m.arena_p]
        assert (enc._momentum, enc._wd, pred._wd) == (0.9, 1e-4, 1e-4)
        assert eng.lr_decay_unit == 'epoch' and type(eng.lr_scheduler).__name__ == 'TimmCosine'
        assert eng.lr_scheduler.T_max == 10 and eng.lr_scheduler.last_epoch == 0
        assert opt.get_lr(0) == 0.1 and opt.get_lr(1) == 0.1
        eng.lr_scheduler.step(5)
        assert abs(opt.get_lr(0) - 0.05) < 1e-12 and opt.get_lr(1) == 0.1        # the predictor's rate is fixed
        ps = dict(m.named_parameters())
        assert not ps['encoder.fc.6.bias'].requires_grad and ps['encoder.fc.6.weight'].requires_grad
        sd = m.state_dict()
        assert 'encoder.fc.7._mean' in sd and 'encoder.fc.7.weight' not in sd and 'predictor.3.bias' in sd
        assert float(sd['encoder.layer3.2.bn3.weight'].abs().max()) == 0.0           # zero_init_residual
    finally:
        hip_config.set_compute_dtype(prev)


@pytest.mark.skipif(not os.path.isdir('/root/reference/tasks/ssl/simsiam'), reason='reference tree not present')
@pytest.mark.parametrize('yaml_path, model, opt_name, lr, compute', [
    ('/root/reference/tasks/ssl/simsiam/configs/simsiam_resnet50_lp_in1k_1n8c_dp_fp32.yaml',
     'SimSiamLinearProbe', 'MomentumLARC', 1.6, torch.float32),
    ('/root/reference/tasks/ssl/mocov3/configs/mocov3_vit_base_patch16_224_lp_in1k_1n8c_dp_fp16o1.yaml',
     'MoCoV3LinearProbe', 'Momentum', 12.0, torch.bfloat16)])
def test_v2_engine_builds_from_the_reference_linear_probe_yamls_unchanged(yaml_path, model, opt_name, lr, compute):
    """The linear-probe recipes of tasks/ssl/{simsiam,mocov3} read as they are (`-o`: device, dataset class, no
    pre-trained file, a smaller run): task_type Classification -> ClassificationTrainingEpochLoop +
    ClassificationEvaluationLoop, Loss / Metric blocks -> CombinedLoss(CELoss) / CombinedMetrics(TopkAcc), the
    probe model with ONLY its classifier trainable, the optimizer of the yaml over that one arena, TimmCosine per
    epoch starting at the base rate (last_epoch: 0), train and evaluation loaders."""
    from passl_amd.engine.engine import Engine
    from passl_amd.hip import config as hip_config
    prev = hip_config.get_compute_dtype()
    try:
        cfg = get_config(yaml_path, ['Global.device=cpu', 'Global.epochs=9', 'Global.pretrained_model=None',
                                     'DataLoader.Train.dataset.name=SyntheticLabeled',
                                     'DataLoader.Eval.dataset.name=SyntheticLabeled',
                                     'DataLoader.Train.sampler.batch_size=4', 'DataLoader.Eval.sampler.batch_size=4'])
        for part in ('Train', 'Eval'):
            cfg.DataLoader[part].dataset.num_samples = 10      # (not keys of the yaml: `-o` cannot add them)
            cfg.DataLoader[part].dataset.image_size = 32 if 'simsiam' in yaml_path else 224
        eng = Engine(cfg, mode='train')
        assert hip_config.get_compute_dtype() == compute
        m, opt = eng.model, eng.optimizer
        assert type(m).__name__ == model and type(opt).__name__ == opt_name
        trainable = [n for n, p in m.named_parameters() if p.requires_grad]
        head = 'fc' if 'simsiam' in yaml_path else 'head'
        assert trainable == [head + '.weight', head + '.bias'] and opt._arenas == [m.arena_q]
        assert m.arena_q.n_train == (2048 if head == 'fc' else 768) * 1000 + 1000
        assert eng.lr_decay_unit == 'epoch' and type(eng.lr_scheduler).__name__ == 'TimmCosine'
        assert eng.lr_scheduler.T_max == 9 and eng.lr_scheduler.last_epoch == 0 and opt.get_lr() == lr
        eng.lr_scheduler.step(3)
        assert abs(opt.get_lr() - 0.5 * lr * (1 + math.cos(math.pi * 3 / 9))) < 1e-12
        if opt_name == 'MomentumLARC':
            assert (opt._momentum, opt._wd, opt._coeff, opt._clip, opt._eps) == (0.9, 0.0, 0.001, False, 1e-8)
        else:
            assert (opt._momentum, opt._wd) == (0.9, 0.0)
        assert type(eng.train_loop).__name__ == 'ClassificationTrainingEpochLoop'
        assert type(eng.validate_loop).__name__ == 'ClassificationEvaluationLoop' and eng.train_loop.val_loop is eng.validate_loop
        assert type(eng.train_loss_func.loss_func[0]).__name__ == 'CELoss' and eng.train_loss_func.loss_weight == [1.0]
        assert eng.eval_metric_func.metric_func_list[0].topk == [1, 5]
        # train: 10 samples / 4 with drop_last False -> 3 batches, the last one with 2 rows; [data, label] batches
        sizes = [(b[0].shape[0], b[1].shape[0]) for b in eng.eval_dataloader]
        assert sizes == [(4, 4), (4, 4), (2, 2)] and len(eng.eval_dataloader.dataset) == 10
        sd = m.state_dict()
        assert (head + '.weight') in sd and abs(float(sd[head + '.weight'].std()) - 0.01) < 1e-3
        assert float(sd[head + '.bias'].abs().max()) == 0.0
    finally:
        hip_config.set_compute_dtype(prev)


def test_no_model_leaves_a_parameter_uninitialised(monkeypatch):
    """hip.nn.Linear / Conv2D allocate their weights with torch.empty and leave the values to the model's init rule
    (the reference's layers draw a framework default first).  With torch.empty poisoned to NaN every parameter and
    buffer of every buildable model must come out finite — the v2 ResNet fc and the SimSiam MLPs, which have no init
    rule of their own in the reference, take Paddle's nn.Linear default (Xavier uniform, zero bias)."""
    from passl_amd.hip import config as hip_config
    from passl_amd.models import build_model
    from passl_amd.modeling.architectures import build_model as build_v110
    hip_config.set_device('cpu')
    real_empty = torch.empty

    def poisoned(*a, **k):
        t = real_empty(*a, **k)
        if t.is_floating_point():
            t.fill_(float('nan'))
        return t
    monkeypatch.setattr(torch, 'empty', poisoned)
    torch.manual_seed(0)
    built = []
    for cfg in (dict(name='simsiam_resnet50_pretrain'), dict(name='simsiam_resnet50_linearprobe', class_num=16),
                dict(name='mocov3_vit_base_pretrain'), dict(name='mocov3_vit_base_linearprobe', class_num=16),
                dict(name='moco_v2_resnet50', K=256), dict(name='simclr_resnet50')):
        built.append((cfg['name'], build_model(cfg)))
    for path in ('configs/mae/mae_vit_b_synthetic.yaml', 'configs/clip/vit-b-32_synthetic.yaml',
                 'configs/moco/moco_clas_r50_synthetic.yaml'):
        cfg = get_config(os.path.join(ROOT, path), [])
        built.append((path, build_v110(cfg.model)))
    for name, m in built:
        bad = [n for n, t in list(m.named_parameters()) + list(m.named_buffers())
               if t.is_floating_point() and not bool(torch.isfinite(t).all())]
        assert not bad, (name, bad[:8])
    sim = dict(built)['simsiam_resnet50_pretrain']
    w = sim.encoder.fc[0].weight
    assert abs(float(w.abs().max()) - math.sqrt(6.0 / 4096)) < 1e-4 and float(sim.predictor[3].bias.abs().max()) == 0.0


def test_trace_timeline_tool_on_a_synthetic_trace(tmp_path):
    """tools/trace_timeline.py (per-stream view of a rocprofv3 --kernel-trace csv): busy time per stream, time with
    k kernels in flight, kernels running alone, idle gaps — on a hand-made two-stream trace with known answers."""
    import subprocess
    import sys
    rows = ['"Kind","Agent_Id","Queue_Id","Stream_Id","Thread_Id","Dispatch_Id","Kernel_Id","Kernel_Name",'
            '"Correlation_Id","Start_Timestamp","End_Timestamp","LDS_Block_Size","Scratch_Size","VGPR_Count",'
            '"Accum_VGPR_Count","SGPR_Count","Workgroup_Size_X","Workgroup_Size_Y","Workgroup_Size_Z","Grid_Size_X",'
            '"Grid_Size_Y","Grid_Size_Z"']

    def k(stream, name, t0, t1, grid=256 * 1024):
        rows.append('"KERNEL_DISPATCH","Agent 2",1,%d,1,1,1,"%s",1,%d,%d,0,0,8,0,16,256,1,1,%d,1,1'
                    % (stream, name, t0, t1, grid))
    # two identical steps of 1 ms: main stream conv 0-600 us, side stream wgrad 200-500 us, a small finalize alone
    # 700-750 us, then the marker 900-1000 us; 100 us + 150 us idle per step
    for s in range(3):
        o = s * 1_000_000
        k(0, 'void conv_kernel<1>(int)', o + 0, o + 600_000)
        k(1, 'void wgrad_kernel(int)', o + 200_000, o + 500_000)
        k(0, 'void bn_finalize_kernel(int)', o + 700_000, o + 750_000, grid=8 * 256)
        k(0, 'void sgd_kernel(float*)', o + 900_000, o + 1_000_000)
    path = tmp_path / 'trace.csv'
    path.write_text('\n'.join(rows) + '\n')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'trace_timeline.py'), str(path), '2'],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    out = r.stdout
    assert 'window: 2 steps, 1.000 ms per step, 4 kernels per step' in out
    assert 'stream 0    busy 0.750 ms per step' in out and 'stream 1    busy 0.300 ms per step' in out
    assert '0 kernels in flight: 0.250 ms per step' in out and '2 kernels in flight: 0.300 ms per step' in out
    assert '1 kernels in flight: 0.450 ms per step' in out
    line = [l for l in out.splitlines() if 'bn_finalize_kernel' in l and '[' in l][0]
    assert '0.050 [0.050]' in line                       # runs alone, and with fewer than 256 workgroups


def test_trace_chain_tool_on_a_synthetic_trace(tmp_path):
    """tools/trace_chain.py (which stream is the critical chain, and what its launches and the gaps in front of them
    cost): on a hand-made two-stream trace with known answers."""
    import subprocess
    import sys
    rows = ['"Kind","Agent_Id","Queue_Id","Stream_Id","Thread_Id","Dispatch_Id","Kernel_Id","Kernel_Name",'
            '"Correlation_Id","Start_Timestamp","End_Timestamp","LDS_Block_Size","Scratch_Size","VGPR_Count",'
            '"Accum_VGPR_Count","SGPR_Count","Workgroup_Size_X","Workgroup_Size_Y","Workgroup_Size_Z","Grid_Size_X",'
            '"Grid_Size_Y","Grid_Size_Z"']

    def k(stream, name, t0, t1):
        rows.append('"KERNEL_DISPATCH","Agent 2",1,%d,1,1,1,"%s",1,%d,%d,0,0,8,0,16,256,1,1,65536,1,1' % (stream, name, t0, t1))
    for s in range(3):                      # steps of 1 ms: conv 0-600, finalize 610-650 (10 us gap), sgd 900-1000 (250 us pause)
        o = s * 1_000_000
        k(0, 'void conv_kernel<1>(int)', o + 0, o + 600_000)
        k(1, 'void wgrad_kernel(int)', o + 200_000, o + 500_000)
        k(0, 'void bn_finalize_kernel(int)', o + 610_000, o + 650_000)
        k(0, 'void sgd_kernel(float*)', o + 900_000, o + 1_000_000)
    path = tmp_path / 'trace.csv'
    path.write_text('\n'.join(rows) + '\n')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'trace_chain.py'), str(path), '2', 'sgd_kernel',
                        '--list', '0'], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    out = r.stdout
    assert 'window: 2 steps, 1.000 ms per step' in out
    # the busiest stream first; a pause of 200 us or more is a wait for another stream, not a launch gap
    first = [l for l in out.splitlines() if l.startswith('stream ')][0]
    assert first.startswith('stream 0: 3 kernels per step, busy 740.0 us, launch gaps (< 200 us each) 10.0 us per step')
    fin = [l for l in out.splitlines() if 'bn_finalize_kernel' in l and 'us' not in l][0].split()
    assert fin[1:] == ['1.0', '40.0', '40.0', '10.0']      # calls, us / step, average us, gap before (us / step)
    assert 'stream 1: 1 kernels per step, busy 300.0 us' in out


def test_step_plan_runs_a_batch_of_another_shape_eagerly(monkeypatch):
    """hip/replay.py:StepPlan.run: a recorded step is shape-specialised; a batch of another shape (the tail batch of a
    `drop_last: False` loader, reference configs/simclr/simclr_r50_IM.yaml:88-90) must run as an eager step, not raise,
    and must not touch the plan (host logic only: the recorded state is faked, nothing is launched)."""
    import torch
    from passl_amd.hip import replay
    calls = []

    def step(x):
        calls.append(tuple(x.shape))
        return {'loss': x.sum()}

    plan = replay.StepPlan(step)
    monkeypatch.setattr(replay, 'plans_enabled', lambda: True)
    monkeypatch.setattr(replay.L, 'stream', lambda: 0)
    plan.handle, plan._main = object(), 0                  # "recorded" for batches of 4 x 3
    plan.static_in, plan._src = [torch.zeros(4, 3)], [None]
    plan.__class__.__del__ = lambda self: None             # the fake handle must not reach plan_destroy
    out = plan.run(torch.ones(2, 3))
    assert calls == [(2, 3)] and float(out['loss']) == 6.0
    assert plan.eager_fallbacks == 1 and plan.replays == 0 and plan.handle is not None
    assert plan._fits([torch.zeros(4, 3)]) and not plan._fits([torch.zeros(4, 3, dtype=torch.float64)])
    plan.handle = None


def test_raw_parameter_readiness_counts_forward_uses():
    """hip/nn.py: class / position embeddings, tokens and logit_scale report their gradient through the same use
    counter as layer parameters — a Function whose forward ran twice in one step (a trunk run once per view) releases
    the all-reduce bucket only with the second backward; without a counted forward the first report releases it."""
    import types
    from types import SimpleNamespace
    import torch
    from passl_amd.hip import nn
    marks = []
    fake = SimpleNamespace(reducer=SimpleNamespace(mark_ready=marks.append), _uses={})
    for name in ('expect_grad', 'grad_ready', 'param_grad_ready'):
        setattr(fake, name, types.MethodType(getattr(nn.EncoderArena, name), fake))
    p = torch.nn.Parameter(torch.zeros(3))
    p._passl_arena, p._passl_index = fake, 5
    nn.param_expect_grad(p)
    nn.param_expect_grad(p)                      # two forward uses in this step
    nn.param_grad_ready(p)
    assert marks == []
    nn.param_grad_ready(p)
    assert marks == [5] and fake._uses == {}
    nn.param_grad_ready(p)                       # a backward without a counted forward: ready at once, as before
    assert marks == [5, 5]
    with torch.no_grad():
        nn.param_expect_grad(p)                  # no backward will come: nothing to wait for
    assert fake._uses == {}


def test_vit_factories_outside_the_attention_envelope_refuse_at_construction():
    """passl.models.vision_transformer: the reference's factories are shape-generic (vision_transformer.py:142-156);
    the HIP attention kernels cover head dimension 32 / 64 and at most 208 tokens.  A factory outside that envelope
    must raise when the model is BUILT (not at its first forward); the 224^2 base / large ones build."""
    import pytest
    from passl_amd.hip import config
    config.set_device('cpu')
    from passl_amd.models import vision_transformer as V
    for name in ('ViT_base_patch16_384', 'ViT_large_patch16_384', 'ViT_huge_patch14_224', 'ViT_g_patch14_224'):
        with pytest.raises(NotImplementedError, match='attention'):
            getattr(V, name)(class_num=0)
    m = V.VisionTransformer(patch_size=32, embed_dim=128, depth=1, num_heads=4, class_num=0)      # ViT-*/32 @ 224: 50 tokens
    assert m.pos_embed.shape[1] == 50


def test_iter_loader_average_meter_and_use_amp_mapping(tmp_path):
    """Verdict r05 hygiene: IterLoader restated (endless, epoch = exhausted passes), AverageMeter.total has its
    postfix, ``use_amp: True`` maps to the bf16 compute dtype instead of raising (trainer.py:186-215)."""
    from passl_amd.engine.trainer import IterLoader, Trainer
    from passl_amd.hip import config as hip_config
    from passl_amd.utils.misc import AverageMeter
    it = IterLoader([10, 11, 12], epoch=4)
    seen = [(next(it), it.epoch) for _ in range(7)]
    assert seen == [(10, 4), (11, 4), (12, 4), (10, 5), (11, 5), (12, 5), (10, 6)]
    assert len(it) == 3
    with pytest.raises(RuntimeError):
        next(IterLoader([]))
    m = AverageMeter('batch_cost', '.3f', postfix=' s')
    m.update(0.5, 2)
    m.update(1.0)
    assert m.total == 'batch_cost_sum: 2.000 s' and str(m) == 'batch_cost: 1.000 (0.667)'
    _register_dummies()
    before = hip_config.get_compute_dtype()
    try:
        hip_config.set_compute_dtype('fp32')
        cfg = get_config(OUR_CFG, ['device=cpu', 'epochs=1', 'dataloader.train.sampler.batch_size=4',
                                   'dataloader.train.dataset.num_samples=4',
                                   'dataloader.train.dataset.image_size=8'])
        cfg.use_amp = True                       # as configs/*/...yaml spell it (top-level key)
        cfg.model = AttrDict(name='DummySSL', dim=4)
        cfg.optimizer = AttrDict(name='PlainSGD')
        cfg.output_dir = str(tmp_path)
        tr = Trainer(cfg)
        assert tr.use_amp is True and tr.scaler is None
        assert hip_config.get_compute_dtype() == torch.bfloat16
    finally:
        hip_config.set_compute_dtype(before)


REF_SIMCLR_R18_CFG = '/root/reference/configs/simclr/simclr_r18_cifar10.yaml'


@pytest.mark.skipif(not os.path.exists(REF_SIMCLR_R18_CFG), reason='reference tree not present')
def test_reference_simclr_r18_cifar10_config_loads_and_builds_unchanged(tmp_path):
    """configs/simclr/simclr_r18_cifar10.yaml (round-5 verdict, missing #1): backbone ``ResNetCifar`` depth 18
    (BasicBlock trunk of resnetcifar.py:41-118, 216-333), ``frozen_stages: 4``, the fc3 neck at 512 channels, lr block
    ``CosineWarmup`` with the SimCLR key set, LARS.  Built from the YAML as written; only the dataset NAME is replaced."""
    hip_config.set_device('cpu')
    from passl_amd.engine.trainer import Trainer
    from passl_amd.modeling import build_model
    from passl_amd.modeling.backbones.resnet import BasicBlock
    from oracle.simclr import init_encoder_state
    cfg = get_config(REF_SIMCLR_R18_CFG, ['dataloader.train.dataset.name=SyntheticCIFAR10'])
    model = build_model(cfg.model)
    bb = model.backbone
    assert type(bb).__name__ == 'ResNetCifar' and bb.frozen_stages == 4 and bb.fully_frozen and bb.with_pool
    assert not hasattr(bb, 'maxpool') and all(isinstance(b, BasicBlock) for st in (bb.layer1, bb.layer4) for b in st)
    assert [len(getattr(bb, 'layer%d' % i)) for i in (1, 2, 3, 4)] == [2, 2, 2, 2]
    assert model.head.temperature == 0.5
    # the reference's own state_dict layout for this trunk (keys and shapes as its ResNetsimclr(depth=18) + fc3 give them)
    ost = init_encoder_state(torch.Generator().manual_seed(0), 512, 512, depth=18)
    sd = model.encoder.state_dict()
    assert list(sd.keys()) == list(ost.keys()) and all(tuple(sd[k].shape) == tuple(ost[k].shape) for k in ost)
    assert sum(p.numel() for p in model.parameters()) == 11769792
    assert sum(p.numel() for p in model.parameters() if p.requires_grad) == 593280      # the projector only
    # the whole Trainer from the YAML: lr block resolved like the r50 recipe's (builder.py:54-66) at the real global batch
    cfg.device = 'cpu'
    cfg.output_dir = str(tmp_path)
    cfg.timestamp = ''
    tr = Trainer(cfg)
    assert tr.iters_per_epoch == 50000 // 512 and 'lars' in tr.optimizer.type
    s = tr.lr_scheduler
    assert type(s).__name__ == 'simclrCosineWarmup'
    assert s.warmup_steps == 10 * 50000 // 512 and s.end_lr == 1.0 * 512 / 256.
    assert s.learning_rate.T_max == 50000 * 1000 // 512 + 1 - s.warmup_steps
    assert tr.optimizer._wd == 1e-4 and tr.optimizer._momentum == 0.9
    with pytest.raises(NotImplementedError):
        from passl_amd.datasets.builder import DATASETS
        DATASETS.get('CIFAR10')(dataroot='data/cifar10/train')


def test_cosine_warmup_scheduler_known_answers():
    """``CosineWarmup`` / ``Cosine`` with their own argument list (passl_v110/solver/lr_scheduler.py:29-102)."""
    from passl_amd.solver import build_lr_scheduler
    cfg = AttrDict(name='CosineWarmup', learning_rate=0.5, warmup_steps=4, start_lr=0.0, end_lr=0.5, T_max=20)
    s = build_lr_scheduler(cfg, 7)
    seen = []
    for _ in range(8):
        seen.append(s())
        s.step()
    assert seen[:4] == [0.0, 0.125, 0.25, 0.375]
    # after the warm-up: the wrapped Cosine stepped to (t - warmup), period T_max - warmup_steps
    assert abs(seen[4] - 0.5) < 1e-12
    for t in (5, 6, 7):
        assert abs(seen[t] - 0.25 * (1 + math.cos(math.pi * (t - 4) / 16))) < 1e-12
    with pytest.raises(ValueError):
        build_lr_scheduler(AttrDict(name='CosineWarmup', total_images=10, warmup_epochs=1, start_lr=0, end_lr=1.0,
                                    T_max=2, learning_rate_scaling='linear'), 7)


def test_host_ring_loader_cycles_its_pinned_batches_on_cpu():
    from passl_amd.datasets.synthetic import HostRingLoader, SyntheticLoader, SyntheticTwoView
    ds = SyntheticTwoView(num_samples=64, image_size=8, seed=5)
    inner = SyntheticLoader(ds, batch_size=4, device=torch.device('cpu'))
    ring = HostRingLoader(inner, ring=3)
    got = [ring.take() for _ in range(7)]
    assert len(ring) == len(inner) == 16 and ring.bytes_per_batch == 2 * 4 * 3 * 8 * 8 * 4
    assert all(torch.equal(got[i][0], got[i + 3][0]) and torch.equal(got[i][1], got[i + 3][1]) for i in range(4))
    assert not torch.equal(got[0][0], got[1][0]) and not torch.equal(got[1][0], got[2][0])
    assert torch.equal(got[0][0], inner._cache[0][0])        # batch 0 = the resident loader's batch (same seed)
    assert sum(1 for _ in ring) == 16


REF_MAE_FT_CFG = '/root/reference/configs/mae/mae_vit_b_finetune.yaml'


@pytest.mark.skipif(not os.path.exists(REF_MAE_FT_CFG), reason='reference tree not present')
def test_reference_mae_finetune_config_loads_and_builds_unchanged():
    """configs/mae/mae_vit_b_finetune.yaml (round-5 verdict, missing #2): MAE_FINETUNE over MAE_ViT (class token,
    learnable position table, global average pool + fc_norm) and VisionTransformerClsHead; LinearWarmup over
    CosineAnnealingDecay; AdamW.  State layout = what the reference's own classes give (tests/golden/mae_ft_vit_b.npz
    records the key list of its state_dict)."""
    hip_config.set_device('cpu')
    from passl_amd.modeling import build_model
    from passl_amd.solver import build_lr_scheduler, build_optimizer
    cfg = get_config(REF_MAE_FT_CFG, [])
    model = build_model(cfg.model)
    assert type(model).__name__ == 'MAE_FINETUNE' and type(model.backbone).__name__ == 'MAE_ViT'
    assert model.backbone.global_pool and not hasattr(model.backbone, 'norm') and hasattr(model.backbone, 'fc_norm')
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'mae_ft_vit_b.npz'))
    want = [str(k) for k in z['keys']]
    got = ['%s:%s' % (k, 'x'.join(map(str, v.shape))) for k, v in model.state_dict().items()]
    assert got == want
    assert sum(p.numel() for p in model.parameters() if p.requires_grad) == 86567656
    sched = build_lr_scheduler(cfg.lr_scheduler, 10)
    assert sched.warmup_steps == 50 and sched.learning_rate.T_max == 1000 and abs(sched() - 1e-6) < 1e-12
    opt = build_optimizer(cfg.optimizer, sched, [model])
    assert opt.type == 'adamw' and (opt._b1, opt._b2, opt._wd) == (0.9, 0.999, 0.05)


def test_step_ab_variant_grammar():
    """tools/step_ab.py (same-box A/B of whole steps): `label:` opens a variant, NAME=value goes to its environment,
    everything else to bench.py's command line."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('step_ab', os.path.join(ROOT, 'tools', 'step_ab.py'))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    v = m.parse_variants(['base:', 'grid512:', 'PASSL_WGRAD_TARGET_BLOCKS=512', 'X=a=b', 'dp:', '--dp-force', '--batch', '128'])
    assert [x['label'] for x in v] == ['base', 'grid512', 'dp']
    assert v[0] == {'label': 'base', 'env': {}, 'flags': []}
    assert v[1]['env'] == {'PASSL_WGRAD_TARGET_BLOCKS': '512', 'X': 'a=b'} and v[1]['flags'] == []
    assert v[2]['flags'] == ['--dp-force', '--batch', '128'] and v[2]['env'] == {}
    with pytest.raises(SystemExit):
        m.parse_variants(['A=1'])
