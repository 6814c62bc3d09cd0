"""v2 ResNet — reference passl/models/resnet.py:52-93: ``ResNet(block, depth=50, width=64, class_num=1000,
with_pool=True, groups=1, zero_init_residual=True)`` = paddle.vision's ResNet (the tree's own statement of it:
passl_v110/modeling/backbones/resnetimagenet.py:111-253) + zero-initialised last BatchNorm of every residual branch,
+ ``Model``'s ``load_pretrained`` / ``save``.  state_dict keys: ``conv1 / bn1 / layer{1..4}.{i}.{conv,bn}{1,2,3} /
downsample.{0,1} / fc``.

The trunk IS the hot path's trunk (passl_amd/modeling/backbones/resnet.py: NHWC implicit-GEMM convs with fused
BatchNorm statistics, streaming BatchNorm kernels, fused max-pool); this class adds the average pool -> flatten -> fc
tail (fc = the GEMM kernel with a bias epilogue and fp32 output).  Bottleneck blocks, width 64, groups 1 only."""
import math
import os
import pickle

import torch

from ..hip import nn as hnn
from ..modeling.backbones.resnet import BottleneckBlock, ResNet as _Trunk
from ..utils.checkpoint import load_lenient, load_pickle, to_numpy
from .base_model import Model

__all__ = ['ResNet', 'BottleneckBlock', 'resnet50']


@torch.no_grad()
def paddle_default_linear_init_(lin):
    """paddle.nn.Linear without a weight_attr: the framework default initializer = Xavier uniform over [in, out]
    (U(-a, a), a = sqrt(6 / (in + out))), zero bias  [Paddle-semantics] — hip.nn.Linear leaves its weight unset for
    the model's own init rule to fill, and the v2 ResNet fc / SimSiam MLPs have none of their own."""
    a = math.sqrt(6.0 / float(lin.in_features + lin.out_features))
    lin.weight.copy_((torch.rand(lin.weight.shape) * 2 - 1) * a)
    if lin.bias is not None:
        lin.bias.zero_()
    return lin


class ResNet(_Trunk, Model):
    def __init__(self, block=BottleneckBlock, depth=50, width=64, class_num=1000, with_pool=True, groups=1,
                 zero_init_residual=True):
        if block is not BottleneckBlock:
            raise NotImplementedError('the HIP path builds bottleneck ResNets (depth 50 / 101 / 152)')
        if width != 64 or groups != 1:
            raise NotImplementedError('wide / grouped ResNets are outside the hot path')
        if class_num > 0 and not with_pool:
            raise NotImplementedError('a fc on an un-pooled feature map')
        _Trunk.__init__(self, depth, num_classes=0, with_pool=with_pool, zero_init_residual=zero_init_residual)
        self.class_num = class_num
        if class_num > 0:
            self.fc = paddle_default_linear_init_(hnn.Linear(512 * BottleneckBlock.expansion, class_num))

    def forward(self, x):
        y = _Trunk.forward(self, x)                      # NHWC; [N, 1, 1, 2048] behind the pool
        if self.class_num > 0:
            y = self.fc(y.reshape(y.shape[0], -1), out_f32=True)
        return y

    def load_pretrained(self, path, rank=0, finetune=False):
        if not os.path.exists(path + '.pdparams'):
            raise ValueError('Model pretrain path {} does not exists.'.format(path))
        load_lenient(self, load_pickle(path + '.pdparams'), what='pretrained model')

    def save(self, path, local_rank=0, rank=0):
        if rank != 0:
            return
        os.makedirs(os.path.dirname(path) or '.', exist_ok=True)
        with open(path + '.pdparams', 'wb') as f:
            pickle.dump(to_numpy(dict(self.state_dict())), f, protocol=2)


def resnet50(**kwargs):
    return ResNet(block=BottleneckBlock, depth=50, **kwargs)
