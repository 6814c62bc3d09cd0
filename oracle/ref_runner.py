"""Run the reference's *own* MoCo-v2 sources on torch-CPU through the paddle
shim.  TEST INFRASTRUCTURE ONLY; needs /root/reference (build container only —
nothing on the GPU box imports this module).

The reference files are imported where they lie (never copied).  Package
``__init__`` files are bypassed (they import every backbone / dataset in the
tree, most of which need far more of Paddle than the shim offers): empty
package objects with the right ``__path__`` are pre-seeded in ``sys.modules`` and
only the files on the hot path are executed:

    utils/registry.py  modules/init.py  modules/freeze.py
    modeling/backbones/{builder,resnetimagenet,resnet}.py
    modeling/necks/{builder,base_neck}.py
    modeling/heads/{builder,contrastive_head}.py
    modeling/architectures/{builder,moco}.py
"""
import importlib
import os
import sys
import types

REF_ROOT = os.environ.get('PASSL_REFERENCE', '/root/reference')
PKG = 'refpassl'


def available():
    return os.path.isdir(os.path.join(REF_ROOT, 'passl_v110'))


def _pkg(name, path):
    m = types.ModuleType(name)
    m.__path__ = [path]
    m.__package__ = name
    sys.modules[name] = m
    return m


def load():
    """Returns a namespace with the reference's MoCo, ContrastiveHead, ... classes."""
    if PKG + '.modeling.architectures.moco' in sys.modules:
        return _namespace()
    from . import paddle_shim
    paddle_shim.install()
    base = os.path.join(REF_ROOT, 'passl_v110')
    _pkg(PKG, base)
    for sub in ('utils', 'modules', 'modeling', 'modeling/backbones', 'modeling/necks',
                'modeling/heads', 'modeling/architectures'):
        _pkg(PKG + '.' + sub.replace('/', '.'), os.path.join(base, sub))
    imp = importlib.import_module
    # utils.logger needs paddle.distributed.ParallelEnv (shimmed)
    imp(PKG + '.utils.registry')
    imp(PKG + '.utils.logger')
    # modules: `from ...modules import freeze_batchnorm_statictis`, `init`, `freeze`
    modules = sys.modules[PKG + '.modules']
    modules.init = imp(PKG + '.modules.init')
    modules.freeze = imp(PKG + '.modules.freeze')
    modules.freeze_batchnorm_statictis = modules.freeze.freeze_batchnorm_statictis
    # backbones: vendored paddle.vision ResNet first, then bind it as paddle.vision.models.ResNet
    rin = imp(PKG + '.modeling.backbones.resnetimagenet')
    paddle_shim.bind_vision_resnet(rin)
    bb = sys.modules[PKG + '.modeling.backbones']
    bb.build_backbone = imp(PKG + '.modeling.backbones.builder').build_backbone
    imp(PKG + '.modeling.backbones.resnet')
    # architectures/builder.py:18 imports discrete_vae names it never uses on this path
    dv = types.ModuleType(PKG + '.modeling.backbones.discrete_vae')
    for n in ('Dalle_VAE', 'DiscreteVAE', 'load_model', 'Encoder', 'Decoder'):
        setattr(dv, n, None)
    sys.modules[dv.__name__] = dv
    nk = sys.modules[PKG + '.modeling.necks']
    nk.build_neck = imp(PKG + '.modeling.necks.builder').build_neck
    imp(PKG + '.modeling.necks.base_neck')
    hd = sys.modules[PKG + '.modeling.heads']
    hd.build_head = imp(PKG + '.modeling.heads.builder').build_head
    imp(PKG + '.modeling.heads.contrastive_head')
    imp(PKG + '.modeling.architectures.builder')
    imp(PKG + '.modeling.architectures.moco')
    # SimCLR row: resnetcifar/resnetsimclr backbone, SimCLRContrastiveHead, SimCLR architecture
    imp(PKG + '.modeling.backbones.resnetcifar')
    imp(PKG + '.modeling.backbones.resnetsimclr')
    imp(PKG + '.modeling.heads.simclr_contrastive_head')
    imp(PKG + '.modeling.architectures.simclr')
    # MAE row: backbones/mae.py (class MAE) over modules/get_sincos_pe.py, architectures/MAE.py.
    # get_sincos_pe.py:25 uses the `np.float` alias that NumPy >= 1.24 removed
    import numpy as _np
    if not hasattr(_np, 'float'):
        _np.float = float
    imp(PKG + '.modules.get_sincos_pe')
    imp(PKG + '.modeling.backbones.mae')
    imp(PKG + '.modeling.architectures.MAE')
    # CLIP row: backbones/{base_transformer,vision_transformer,clip}.py, heads/clip_head.py,
    # architectures/CLIPWrapper.py
    imp(PKG + '.modeling.backbones.base_transformer')
    imp(PKG + '.modeling.backbones.vision_transformer')
    imp(PKG + '.modeling.backbones.clip')
    imp(PKG + '.modeling.heads.clip_head')
    imp(PKG + '.modeling.architectures.CLIPWrapper')
    # linear-probe row: heads/clas_head.py, architectures/clas.py
    imp(PKG + '.modeling.heads.clas_head')
    imp(PKG + '.modeling.architectures.clas')
    # MAE fine-tuning row: heads/vision_transformer_head.py (MAE_FINETUNE / MAE_ViT sit in files loaded above)
    imp(PKG + '.modeling.heads.vision_transformer_head')
    return _namespace()


def _namespace():
    ns = types.SimpleNamespace()
    ns.moco = sys.modules[PKG + '.modeling.architectures.moco']
    ns.MoCo = ns.moco.MoCo
    ns.ContrastiveHead = sys.modules[PKG + '.modeling.heads.contrastive_head'].ContrastiveHead
    ns.accuracy = sys.modules[PKG + '.modeling.heads.contrastive_head'].accuracy
    ns.build_model = sys.modules[PKG + '.modeling.architectures.builder'].build_model
    ns.registry = sys.modules[PKG + '.utils.registry']
    ns.MODELS = sys.modules[PKG + '.modeling.architectures.builder'].MODELS
    ns.BACKBONES = sys.modules[PKG + '.modeling.backbones.builder'].BACKBONES
    ns.NECKS = sys.modules[PKG + '.modeling.necks.builder'].NECKS
    ns.HEADS = sys.modules[PKG + '.modeling.heads.builder'].HEADS
    ns.mae = sys.modules[PKG + '.modeling.backbones.mae']
    ns.MAE = ns.mae.MAE
    ns.MAE_PRETRAIN = sys.modules[PKG + '.modeling.architectures.MAE'].MAE_PRETRAIN
    ns.CLIP = sys.modules[PKG + '.modeling.backbones.clip'].CLIP
    ns.CLIPWrapper = sys.modules[PKG + '.modeling.architectures.CLIPWrapper'].CLIPWrapper
    ns.SimCLR = sys.modules[PKG + '.modeling.architectures.simclr'].SimCLR
    ns.SimCLRContrastiveHead = sys.modules[
        PKG + '.modeling.heads.simclr_contrastive_head'].SimCLRContrastiveHead
    return ns


MOCO_V2_CFG = dict(
    name='MoCo',
    backbone=dict(name='ResNet', depth=50),
    neck=dict(name='NonLinearNeckV1', in_channels=2048, hid_channels=2048,
              out_channels=128, with_avg_pool=True),
    head=dict(name='ContrastiveHead', temperature=0.2),
)


MOCO_V1_NECK = dict(name='LinearNeck', in_channels=2048, out_channels=128, with_avg_pool=True)


def build_reference_moco(K=65536, dim=128, m=0.999, T=0.2, neck='NonLinearNeckV1'):
    """MoCo built by the reference's registries from the configs/moco/moco_v2_r50.yaml
    `model:` block (restated in MOCO_V2_CFG; T=0.2 there is the head temperature,
    the architecture's own T default is unused by train_iter)."""
    import copy
    ns = load()
    cfg = copy.deepcopy(MOCO_V2_CFG)
    cfg.update(K=K, dim=dim, m=m)
    cfg['head']['temperature'] = T
    if neck == 'LinearNeck':              # configs/moco/moco_v1_r50.yaml `model:` block
        cfg['neck'] = copy.deepcopy(MOCO_V1_NECK)
    return ns.build_model(cfg)


def load_oracle_state(model, oracle):
    """Copy a MoCoOracle's q/k/queue state into a reference MoCo instance."""
    import torch
    with torch.no_grad():
        for enc, st in (('encoder_q', oracle.q), ('encoder_k', oracle.k)):
            sd = getattr(model, enc).state_dict()
            assert list(sd.keys()) == list(st.keys()), 'state_dict key order differs'
            for n, t in st.items():
                assert sd[n].shape == t.shape, (n, sd[n].shape, t.shape)
                sd[n].copy_(t.detach())
        model.queue.copy_(oracle.queue)
        model.queue_ptr[0] = oracle.queue_ptr


SIMCLR_CFG = dict(
    name='SimCLR',
    backbone=dict(name='ResNetsimclr', depth=50),
    neck=dict(name='NonLinearNeckfc3', in_channels=2048, hid_channels=2048, out_channels=128,
              with_avg_pool=False),
    head=dict(name='SimCLRContrastiveHead', temperature=0.1),
)


def build_reference_simclr(T=0.1):
    """SimCLR built by the reference's registries from the `model:` block of
    configs/simclr/simclr_r50_IM.yaml (restated in SIMCLR_CFG)."""
    import copy
    ns = load()
    cfg = copy.deepcopy(SIMCLR_CFG)
    cfg['head']['temperature'] = T
    return ns.build_model(cfg)


def load_simclr_state(model, oracle):
    """Copy a SimCLROracle's encoder state into a reference SimCLR instance."""
    import torch
    with torch.no_grad():
        sd = model.encoder.state_dict()
        assert list(sd.keys()) == list(oracle.st.keys()), 'state_dict key order differs'
        for n, t in oracle.st.items():
            assert sd[n].shape == t.shape, (n, sd[n].shape, t.shape)
            sd[n].copy_(t.detach())


def build_reference_mae(cfg, norm_pix_loss=False):
    """The reference's MAE backbone (passl_v110/modeling/backbones/mae.py:318) built through its
    BACKBONES registry with the architecture block of configs/mae/mae_vit_b_pretrain.yaml."""
    ns = load()
    arch = dict(name='MAE', img_size=cfg['img_size'], patch_size=cfg['patch_size'],
                embed_dim=cfg['embed_dim'], depth=cfg['depth'], num_heads=cfg['num_heads'],
                decoder_embed_dim=cfg['decoder_embed_dim'], decoder_depth=cfg['decoder_depth'],
                decoder_num_heads=cfg['decoder_num_heads'], mlp_ratio=cfg['mlp_ratio'],
                norm_pix_loss=norm_pix_loss)
    return ns.BACKBONES.get('MAE')(**{k: v for k, v in arch.items() if k != 'name'})


def load_mae_state(model, oracle):
    import torch
    with torch.no_grad():
        sd = model.state_dict()
        assert set(sd.keys()) == set(oracle.st.keys()), set(sd.keys()) ^ set(oracle.st.keys())
        for n, t in oracle.st.items():
            assert sd[n].shape == t.shape, (n, sd[n].shape, t.shape)
            sd[n].copy_(t.detach())


def build_reference_clip(cfg):
    """The reference's CLIPWrapper (architectures/CLIPWrapper.py:27) built by its MODELS registry from
    the `model:` block of configs/clip/vit-b-32.yaml with the sizes in `cfg`."""
    ns = load()
    arch = dict(name='CLIP', qkv_bias=True, pre_norm=True, proj=True, patch_bias=False)
    arch.update({k: cfg[k] for k in ('embed_dim', 'image_resolution', 'vision_layers', 'vision_width',
                                     'vision_patch_size', 'context_length', 'vocab_size',
                                     'transformer_width', 'transformer_heads', 'transformer_layers')})
    return ns.build_model(dict(name='CLIPWrapper', architecture=arch, head=dict(name='CLIPHead')))


def load_clip_state(model, oracle):
    import torch
    with torch.no_grad():
        sd = model.model.state_dict()
        assert set(sd.keys()) == set(oracle.st.keys()), set(sd.keys()) ^ set(oracle.st.keys())
        for n, t in oracle.st.items():
            assert sd[n].shape == t.shape, (n, sd[n].shape, t.shape)
            sd[n].copy_(t.detach())


def build_reference_clas(num_classes=1000, frozen_stages=4):
    """The reference's Classification model built by its MODELS registry from the `model:` block of
    configs/moco/moco_clas_r50.yaml (frozen_stages = 4; 0..3 = partially frozen trunk,
    passl_v110/modeling/backbones/resnet.py:90-106)."""
    ns = load()
    return ns.build_model(dict(name='Classification', backbone=dict(name='ResNet', depth=50,
                                                                    frozen_stages=frozen_stages),
                               head=dict(name='ClasHead', with_avg_pool=True, in_channels=2048,
                                         num_classes=num_classes)))


def load_clas_state(model, oracle):
    import torch
    with torch.no_grad():
        sd = model.state_dict()
        assert set(sd.keys()) == set(oracle.st.keys()), set(sd.keys()) ^ set(oracle.st.keys())
        for n, t in oracle.st.items():
            assert sd[n].shape == t.shape, (n, sd[n].shape, t.shape)
            sd[n].copy_(t.detach())
