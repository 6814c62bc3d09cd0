"""Projector necks on the HIP path.

``NonLinearNeckV1`` (MoCo-v2: avgpool -> fc -> relu -> fc) and ``LinearNeck`` with the reference's
constructor signatures, sub-layer names (``avgpool``, ``mlp.0``, ``mlp.2``, ``fc``) and init
(kaiming-normal fan_in/relu, bias 0) — passl_v110/modeling/necks/base_neck.py:24-97.
Input is the backbone's NHWC feature map; the first Linear fuses bias+ReLU in its epilogue, the
last Linear writes fp32 (the InfoNCE head works in fp32).
"""
import torch

from ...hip import nn
from ...modules.init import kaiming_init, constant_, normal_init
from .builder import NECKS


def _init_parameters(module, init_linear='normal', std=0.01, bias=0.):
    assert init_linear in ['normal', 'kaiming'], 'Undefined init_linear: {}'.format(init_linear)
    for m in module.modules():
        if isinstance(m, nn.Linear):
            if init_linear == 'normal':
                normal_init(m, std=std, bias=bias)
            else:
                kaiming_init(m, mode='fan_in', nonlinearity='relu')
        elif isinstance(m, nn._BatchNormBase):
            constant_(m.weight, 1)
            constant_(m.bias, 0)


@NECKS.register()
class LinearNeck(nn.Layer):
    def __init__(self, in_channels, out_channels, with_avg_pool=True):
        super().__init__()
        self.with_avg_pool = with_avg_pool
        if with_avg_pool:
            self.avgpool = nn.AdaptiveAvgPool2D((1, 1))
        self.fc = nn.Linear(in_channels, out_channels)
        self.init_parameters()

    def init_parameters(self, init_linear='kaiming'):
        _init_parameters(self, init_linear)

    def forward(self, x):
        if self.with_avg_pool:
            x = self.avgpool(x)
        return self.fc(x.reshape(x.shape[0], -1), out_f32=True)


@NECKS.register()
class NonLinearNeckV1(nn.Layer):
    """The non-linear neck in MoCo v2: fc-relu-fc."""

    def __init__(self, in_channels, hid_channels, out_channels, with_avg_pool=True):
        super().__init__()
        self.with_avg_pool = with_avg_pool
        if with_avg_pool:
            self.avgpool = nn.AdaptiveAvgPool2D((1, 1))
        self.mlp = torch.nn.Sequential(nn.Linear(in_channels, hid_channels), nn.ReLU(),
                                       nn.Linear(hid_channels, out_channels))
        self.init_parameters()

    def init_parameters(self, init_linear='kaiming'):
        _init_parameters(self, init_linear)

    def forward(self, x):
        if self.with_avg_pool:
            x = self.avgpool(x)
        x = x.reshape(x.shape[0], -1)
        x = self.mlp[0](x, relu=True)            # fc + bias + ReLU in one epilogue
        return self.mlp[2](x, out_f32=True)


@NECKS.register()
class NonLinearNeckfc3(nn.Layer):
    """SimCLR projector: fc-BN1D-ReLU-fc-BN1D-ReLU-fc-BN1D, then l2_normalize — reference
    passl_v110/modeling/necks/base_neck.py:209-239 (Linear weights ~ N(0, 0.01),
    modules/init.py:406-412; sub-layer names mlp.0 ... mlp.7).  BatchNorm+ReLU run as the
    training-mode BN kernels over [N, C] rows.  Mixed precision: every Linear accumulates and
    WRITES fp32, the three BatchNorm1D layers work on fp32 rows (a batch-statistics BN over a few
    hundred rows subtracts nearly equal numbers: bf16 inputs would lose most of the signal), and
    only the Linear operands are cast to the compute dtype."""

    def __init__(self, in_channels, hid_channels, out_channels, with_avg_pool=True):
        super().__init__()
        self.with_avg_pool = with_avg_pool
        if with_avg_pool:
            self.avgpool = nn.AdaptiveAvgPool2D((1, 1))
        self.mlp = torch.nn.Sequential(nn.Linear(in_channels, hid_channels),
                                       nn.BatchNorm1D(hid_channels), nn.ReLU(),
                                       nn.Linear(hid_channels, hid_channels),
                                       nn.BatchNorm1D(hid_channels), nn.ReLU(),
                                       nn.Linear(hid_channels, out_channels),
                                       nn.BatchNorm1D(out_channels))
        for m in self.mlp.modules():                       # init_backbone_weight_simclr
            if isinstance(m, nn.Linear):
                normal_init(m, std=0.01, bias=0.)

    def init_parameters(self, init_linear='normal'):
        _init_parameters(self, init_linear)

    def forward(self, x):
        if self.with_avg_pool and x.dim() == 4:
            x = self.avgpool(x)
        x = x.reshape(x.shape[0], -1)                      # layers.squeeze(x, axes=[])
        dt = x.dtype
        x = self.mlp[1](self.mlp[0](x, out_f32=True), relu=True)
        x = self.mlp[4](self.mlp[3](nn.to_compute(x, dt), out_f32=True), relu=True)
        x = self.mlp[7](self.mlp[6](nn.to_compute(x, dt), out_f32=True))
        return nn.normalize(x, axis=1)
