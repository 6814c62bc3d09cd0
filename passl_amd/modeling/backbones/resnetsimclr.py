"""ResNetsimclr: the SimCLR backbone of the reference (passl_v110/modeling/backbones/
resnetsimclr.py:25-91 over resnetcifar.py:216-333) on the HIP path.

Same bottleneck topology and state_dict keys as ``ResNet`` but (1) NO stem max-pool
(resnetcifar.py:275 is commented out; forward :321-333) — ~3.9x the FLOPs of a standard R50 at
224^2; (2) ``with_pool`` defaults to True: the backbone returns the globally pooled [N, 2048]
features; (3) convs initialised XavierNormal(fan_in=None, fan_out=0) = N(0, sqrt(2/fan_in))
(resnetcifar.py:62-70 ...), ``init_parameters()`` is NOT called (resnetsimclr.py:63 comments it
out); BN gamma=1, beta=0.
"""
import math


from ...hip import nn
from ...modules import init
from .builder import BACKBONES
from .resnet import ResNet


@BACKBONES.register()
class ResNetsimclr(ResNet):
    stem_pool = False

    def __init__(self, depth, num_classes=0, with_pool=True, zero_init_residual=False,
                 frozen_stages=-1, pretrained=None):
        super().__init__(depth, num_classes=num_classes, with_pool=with_pool,
                         zero_init_residual=zero_init_residual, frozen_stages=frozen_stages,
                         pretrained=pretrained)

    def init_parameters(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2D):
                fan_in, _ = init._calculate_fan_in_and_fan_out(m.weight)
                init.normal_(m.weight, 0.0, math.sqrt(2.0 / fan_in))     # XavierNormal, fan_out=0
            elif isinstance(m, nn._BatchNormBase):
                init.constant_init(m, 1)


@BACKBONES.register()
class ResNetCifar(ResNetsimclr):
    """``backbone: {name: ResNetCifar, depth: 18, frozen_stages: 4}`` of configs/simclr/simclr_r18_cifar10.yaml:5-8.

    In the reference the name is only an import alias (passl_v110/modeling/backbones/__init__.py:15
    ``from .resnetcifar import ResNet as ResNetCifar``): the class is never registered with BACKBONES — its
    ``__name__`` is ``ResNet`` and resnetcifar.py carries no ``@BACKBONES.register()`` — and its constructor
    ``(block, depth, num_classes, with_pool)`` takes neither the yaml's ``frozen_stages`` nor a depth without a block
    class, so the reference cannot build this config (tests/test_oracle_simclr.py runs its builder on the yaml and
    pins the failure).  What the yaml describes is the class that file is the base of: the no-max-pool trunk of
    resnetcifar.py:216-333 with BasicBlocks (:41-118) behind ResNetsimclr's constructor (depth, frozen_stages, pooled
    [N, 512] output for the ``in_channels: 512`` neck) — registered here under the name the yaml uses."""
