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
