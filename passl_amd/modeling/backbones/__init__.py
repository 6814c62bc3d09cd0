from .builder import BACKBONES, build_backbone
from .resnet import ResNet, BottleneckBlock
from .resnetsimclr import ResNetsimclr
from .mae import MAE
