"""v2 model front end — reference passl/models/__init__.py:37-44 (``build_model(config)``: pop ``name``,
instantiate the factory of that name from this module, require a ``Model``) and
passl/models/base_model.py:25-40 (``Model`` = Layer + ``load_pretrained`` / ``save``).

Thin by design: the factories below build the SAME registry-built architectures the v110 Trainer
builds (passl_amd/modeling/architectures: MoCo, SimCLR, MAE_PRETRAIN, CLIPWrapper) — same HIP
kernels, same EncoderArena storage — and adapt the call convention of the v2 loops
(``model(sub_batch) -> loss tensor | dict``, passl/engine/loops/contrastive_learning_loop.py:52-54).
"""
import copy
import os
import sys

import torch

from ..hip import nn as hnn
from ..modeling.architectures import build_model as _build_v110
from ..utils.checkpoint import load_lenient, load_pickle, to_numpy

__all__ = ['build_model', 'Model']


from .base_model import Model          # noqa: E402  (passl/models/base_model.py)


class ArchModel(Model):
    """A v110 architecture behind the v2 ``Model`` contract.  ``forward(inputs)`` takes the loop's
    sub-batch (a list of tensors, e.g. the two views) and returns the architecture's output dict
    (key 'loss' + logged scalars); parameters / state_dict are the architecture's own."""

    def __init__(self, arch_cfg):
        super().__init__()
        self.arch = _build_v110(copy.deepcopy(arch_cfg))

    def forward(self, inputs, **kw):
        if torch.is_tensor(inputs):
            inputs = [inputs]
        return self.arch(*inputs, mode='train', **kw)

    # state lives in the architecture: expose its keys unprefixed so that v110 checkpoints,
    # tools/extract_weight.py and Model.save files are interchangeable
    def state_dict(self, *a, **k):
        return self.arch.state_dict(*a, **k)

    def load_state_dict(self, sd, strict=True):
        return self.arch.load_state_dict(sd, strict=strict)

    def load_pretrained(self, path, rank=0, finetune=False):
        """``path`` without extension, as in the reference (``<path>.pdparams``)."""
        fn = path if os.path.exists(path) else path + '.pdparams'
        if not os.path.exists(fn):
            raise ValueError('Model pretrain path {} does not exists.'.format(fn))
        sd = load_pickle(fn)
        if 'state_dict' in sd:
            sd = sd['state_dict']
        load_lenient(self.arch, sd, what='pretrained model')

    def save(self, path, local_rank=0, rank=0):
        import pickle
        if rank != 0:
            return
        os.makedirs(os.path.dirname(path) or '.', exist_ok=True)
        with open(path + '.pdparams', 'wb') as f:
            pickle.dump(to_numpy(dict(self.state_dict())), f, protocol=2)


# ---- factories (names follow the reference's `<method>_<backbone>` convention)
def moco_v2_resnet50(dim=128, K=65536, m=0.999, T=0.2, **kw):
    """configs/moco/moco_v2_r50.yaml `model:` block (passl_v110/modeling/architectures/moco.py)."""
    return ArchModel(dict(name='MoCo', backbone=dict(name='ResNet', depth=50),
                          neck=dict(name='NonLinearNeckV1', in_channels=2048, hid_channels=2048,
                                    out_channels=dim, with_avg_pool=True),
                          head=dict(name='ContrastiveHead', temperature=T), dim=dim, K=K, m=m, **kw))


def moco_v1_resnet50(dim=128, K=65536, m=0.999, T=0.07, **kw):
    """configs/moco/moco_v1_r50.yaml."""
    return ArchModel(dict(name='MoCo', backbone=dict(name='ResNet', depth=50),
                          neck=dict(name='LinearNeck', in_channels=2048, out_channels=dim,
                                    with_avg_pool=True),
                          head=dict(name='ContrastiveHead', temperature=T), dim=dim, K=K, m=m, **kw))


def simclr_resnet50(dim=128, T=0.1, multi_rank=False, **kw):
    """configs/simclr/simclr_r50_IM.yaml `model:` block."""
    return ArchModel(dict(name='SimCLR', backbone=dict(name='ResNetsimclr', depth=50),
                          neck=dict(name='NonLinearNeckfc3', in_channels=2048, hid_channels=4096,
                                    out_channels=dim, with_avg_pool=True),
                          head=dict(name='SimCLRContrastiveHead', temperature=T, return_accuracy=True,
                                    multi_rank=multi_rank), dim=dim, T=T, **kw))


from .mocov3 import (MoCoV3ViT, MoCoV3LinearProbe, MoCoV3Pretrain, mocov3_vit_base,      # noqa: E402,F401
                     mocov3_vit_base_linearprobe, mocov3_vit_base_pretrain)
from .resnet import ResNet, resnet50                                   # noqa: E402,F401
from .vision_transformer import *                                     # noqa: E402,F401,F403  (passl/models/__init__.py:24)
from .vision_transformer import VisionTransformer                     # noqa: E402,F401
from .mae import *                                                    # noqa: E402,F401,F403  (passl/models/__init__.py:31)
from .mae import (MaskedAutoencoderViT, mae_vit_base_patch16_dec512d8b, mae_vit_large_patch16_dec512d8b,   # noqa: E402,F401
                  mae_vit_huge_patch14_dec512d8b)
from .simsiam import (SimSiamPretain, SimSiamLinearProbe, simsiam_resnet50_pretrain,      # noqa: E402,F401
                      simsiam_resnet50_linearprobe)


def build_model(config):
    config = copy.deepcopy(dict(config))
    model_type = config.pop('name')
    factory = getattr(sys.modules[__name__], model_type, None)
    if factory is None or not callable(factory):
        raise AttributeError('passl.models has no model named %r' % (model_type,))
    model = factory(**config)
    assert isinstance(model, Model), 'model must inherit from passl.models.Model'
    return model
