"""Run the reference's own v2 MoCo-v3 sources (passl/models/{base_model,vision_transformer,mocov3}.py,
passl/nn/init.py, passl/models/utils/averaged_model.py, passl/utils/{misc,infohub}.py) on torch-CPU through the
paddle shim.  TEST INFRASTRUCTURE ONLY; needs /root/reference (build container only).

The v2 tree imports itself by its absolute package name ``passl``, which in THIS repository is the alias package of
the product (``import passl.models`` is ``passl_amd.models``).  The reference files are therefore loaded in a
process of their own (tests/golden/make_golden_mocov3.py, run as a script) in which ``passl`` is pre-seeded with
empty package objects whose ``__path__`` points into /root/reference — package ``__init__`` files are bypassed, as
in oracle/ref_runner.py — and the product's alias package is never imported.  ``passl.utils.logger`` (log
formatting over paddle.distributed, not part of the algorithm) is replaced by the standard logging module.
"""
import importlib
import logging
import os
import sys
import types

REF_ROOT = os.environ.get('PASSL_REFERENCE', '/root/reference')


def available():
    return os.path.isdir(os.path.join(REF_ROOT, 'passl', 'models'))


def _pkg(name, path):
    m = types.ModuleType(name)
    m.__path__ = [path]
    m.__package__ = name
    sys.modules[name] = m
    return m


def load():
    """-> namespace(mocov3 module, runtime_info_hub).  Refuses to run next to the product's `passl` alias."""
    mod = sys.modules.get('passl')
    if mod is not None and not getattr(mod, '_is_reference_stub', False):
        raise RuntimeError('the product\'s `passl` alias package is already imported in this process: run the '
                           'reference v2 sources in a process of their own')
    from . import paddle_shim
    paddle_shim.install()
    base = os.path.join(REF_ROOT, 'passl')
    root = _pkg('passl', base)
    root._is_reference_stub = True
    for sub in ('models', 'models/utils', 'nn', 'utils'):
        _pkg('passl.' + sub.replace('/', '.'), os.path.join(base, sub))
    lg = types.ModuleType('passl.utils.logger')
    log = logging.getLogger('refpassl')
    for n in ('info', 'warning', 'debug', 'error'):
        setattr(lg, n, getattr(log, n))
    sys.modules['passl.utils.logger'] = lg
    sys.modules['passl.utils'].logger = lg
    imp = importlib.import_module
    sys.modules['passl.nn'].init = imp('passl.nn.init')
    imp('passl.utils.misc')
    hub = imp('passl.utils.infohub')
    imp('passl.models.base_model')
    imp('passl.models.vision_transformer')
    imp('passl.models.utils.averaged_model')
    m3 = imp('passl.models.mocov3')
    return types.SimpleNamespace(mocov3=m3, runtime_info_hub=hub.runtime_info_hub,
                                 vit=sys.modules['passl.models.vision_transformer'])


def load_simsiam():
    """-> namespace(simsiam module).  passl/models/resnet.py subclasses ``paddle.vision.models.resnet.ResNet`` — a
    class of the Paddle wheel (2.4 line), not of the reference tree.  The tree carries its own copy of that class
    (passl_v110/modeling/backbones/resnetimagenet.py, used by the v110 models): the v2 sources are executed against
    THAT copy, adapted to the wheel's constructor signature (width / groups accepted at their defaults only)."""
    ns = load()
    from . import paddle_shim
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        'passl_ref_resnetimagenet', os.path.join(REF_ROOT, 'passl_v110', 'modeling', 'backbones', 'resnetimagenet.py'))
    rin = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(rin)

    class PDResNet(rin.ResNet):
        def __init__(self, block, depth=50, width=64, num_classes=1000, with_pool=True, groups=1):
            assert width == 64 and groups == 1, 'only the plain ResNet of the vendored copy'
            super().__init__(block, depth, num_classes=num_classes, with_pool=with_pool)
    shadow = types.SimpleNamespace(ResNet=PDResNet, BasicBlock=rin.BasicBlock, BottleneckBlock=rin.BottleneckBlock)
    paddle_shim.bind_vision_resnet(shadow)
    importlib.import_module('passl.models.resnet')
    ns.simsiam = importlib.import_module('passl.models.simsiam')
    return ns


def load_mae():
    """-> namespace(+ mae module): passl/models/mae.py (MaskedAutoencoderViT, the mae_vit_* factories,
    MAEVisionTransformer) over passl/models/vision_transformer.py and passl/models/utils/pos_embed.py."""
    ns = load()
    importlib.import_module('passl.models.utils.pos_embed')
    ns.mae = importlib.import_module('passl.models.mae')
    return ns


def _exec_init(name):
    """Run a reference package's own __init__.py inside its pre-seeded package object (relative imports resolve
    through ``__path__``)."""
    m = sys.modules[name]
    fn = os.path.join(m.__path__[0], '__init__.py')
    with open(fn) as f:
        exec(compile(f.read(), fn, 'exec'), m.__dict__)
    return m


def _adamw_op(p, grad, lr, m, v, beta1_pow, beta2_pow, master, p_out, m_out, v_out, b1_out, b2_out, master_out, *attrs):
    """The Paddle `adamw` op as passl/optimizer/adamw.py:124-137 calls it (a STATEMENT of the kernel, not reference code
    [Paddle-semantics]: paddle/phi/kernels/funcs/adam_functors.h + adamw_kernel): in place on p / m / v,
        p *= 1 - lr*coeff (with_decay);  m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;
        p -= lr*sqrt(1-b2^t)/(1-b1^t) * m / (sqrt(v) + eps*sqrt(1-b2^t))
    with b^t = the beta-pow INPUTS.  What IS executed from the reference around it: the per-parameter state, the step
    count, beta^step, the learning rate read (`_get_lr`), with_decay / coeff / multi_precision plumbing."""
    import torch
    a = dict(zip(attrs[0::2], attrs[1::2]))
    b1, b2, eps = a['beta1'], a['beta2'], a['epsilon']
    lr_ = float(lr) * a.get('lr_ratio', 1.0)
    b1p, b2p = float(beta1_pow), float(beta2_pow)
    tgt = master if master is not None else p
    with torch.no_grad():
        g = grad.to(tgt.dtype)
        if a.get('with_decay', False):
            tgt.mul_(1.0 - lr_ * a['coeff'])
        m.mul_(b1).add_(g.to(m.dtype), alpha=1 - b1)
        v.mul_(b2).add_((g * g).to(v.dtype), alpha=1 - b2)
        corr2 = (1.0 - b2p) ** 0.5
        tgt.sub_(lr_ * corr2 / (1.0 - b1p) * m.to(tgt.dtype) / (v.sqrt().to(tgt.dtype) + eps * corr2))
        if master is not None:
            p.copy_(master.to(p.dtype))
    return p, m, v, beta1_pow, beta2_pow, master


def load_solver(ns=None):
    """Adds the v2 solver / loss / metric sources to the namespace: passl/optimizer/{optimizer,momentum,
    momentum_larc}.py (pure-Python update rules: executed as they are), passl/scheduler/lr_scheduler.py (TimmCosine over
    the shim's statement of paddle.optimizer.lr.LRScheduler), passl/loss (CombinedLoss / CELoss) and passl/metric
    (CombinedMetrics / TopkAcc over paddle.metric.accuracy).  passl/optimizer/adamw.py is loaded too, but its update is
    the Paddle kernel ``_C_ops.adamw``: that one call is answered by ``_adamw_op`` (a statement of the kernel)."""
    ns = ns or load()
    import paddle
    if not hasattr(paddle, '_legacy_C_ops'):
        c_ops = types.ModuleType('paddle._legacy_C_ops')       # imported by momentum.py, used only on sparse paths
        c_ops.adamw = _adamw_op
        sys.modules['paddle._legacy_C_ops'] = c_ops
        paddle._legacy_C_ops = c_ops
    base = os.path.join(REF_ROOT, 'passl')
    for sub in ('optimizer', 'scheduler', 'loss', 'metric'):
        if 'passl.' + sub not in sys.modules:
            _pkg('passl.' + sub, os.path.join(base, sub))
    imp = importlib.import_module
    imp('passl.optimizer.optimizer')
    ns.momentum = imp('passl.optimizer.momentum')
    ns.momentum_larc = imp('passl.optimizer.momentum_larc')
    ns.adamw = imp('passl.optimizer.adamw')              # python wrapper executed; the op itself is _adamw_op
    ns.lr_scheduler = imp('passl.scheduler.lr_scheduler')
    ns.loss = _exec_init('passl.loss')
    ns.metric = _exec_init('passl.metric')
    return ns


def _core_stub():
    """passl.core as a package object whose __init__ is bypassed (it imports the fused-parameter machinery over
    paddle.fluid): grad_sync / param_sync are one-rank no-ops, sub-modules (grad_clip) import from the tree."""
    core = sys.modules.get('passl.core')
    if core is None:
        core = _pkg('passl.core', os.path.join(REF_ROOT, 'passl', 'core'))
        core.grad_sync = lambda param_groups, **k: None
        core.param_sync = lambda *a, **k: None
        sys.modules['passl'].core = core
    return core


def load_optimizer_builder(ns):
    """Adds passl/optimizer/__init__.py itself (build_optimizer, group_params, build_group_lr_scheduler) and
    passl/scheduler/__init__.py (build_lr_scheduler).  Stand-ins: passl.core.param_fuse (tensor fusion is skipped off
    the GPU anyway: `'gpu' not in paddle.get_device()`), paddle.optimizer.adam (imported by adan.py)."""
    import paddle
    _core_stub()
    pf = types.ModuleType('passl.core.param_fuse')
    pf.get_fused_params = lambda params: params
    sys.modules['passl.core.param_fuse'] = pf
    sys.modules['passl.core'].param_fuse = pf
    paddle.get_device = lambda: 'cpu'
    adam = types.ModuleType('paddle.optimizer.adam')
    sys.modules['paddle.optimizer.adam'] = adam
    paddle.optimizer.adam = adam
    base = os.path.join(REF_ROOT, 'passl')
    if 'passl.optimizer.utils' not in sys.modules:
        _pkg('passl.optimizer.utils', os.path.join(base, 'optimizer', 'utils'))
    ns.scheduler = _exec_init('passl.scheduler')
    ns.optimizer = _exec_init('passl.optimizer')
    return ns


def load_loops(ns):
    """Adds the v2 loops (passl/engine/loops/{loop,classification_loop}.py) — driven by a stand-in trainer object that
    carries the attributes the loops read (see tests/golden/make_golden_linprobe_v2.py).  Replaced by empty stand-ins:
    passl.utils.io (paddle.save / load plumbing), passl.utils.profiler, passl.core (grad_sync / param_sync: one rank)."""
    base = os.path.join(REF_ROOT, 'passl')
    for sub in ('engine', 'engine/loops'):
        name = 'passl.' + sub.replace('/', '.')
        if name not in sys.modules:
            _pkg(name, os.path.join(base, sub))
    lg = sys.modules['passl.utils.logger']
    lg.dict_format = lambda d, float_placeholders='{:.5f}': ', '.join(
        '{}: {}'.format(k, getattr(v, 'avg', v)) for k, v in d.items())
    lg.scaler = lambda *a, **k: None
    for name in ('passl.utils.io', 'passl.utils.profiler'):
        m = types.ModuleType(name)
        sys.modules[name] = m
        setattr(sys.modules['passl.utils'], name.rsplit('.', 1)[1], m)
    core = _core_stub()
    importlib.import_module('passl.engine.loops.loop')
    ns.classification_loop = importlib.import_module('passl.engine.loops.classification_loop')
    return ns
