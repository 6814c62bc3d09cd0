from .builder import BACKBONES, build_backbone
from .resnet import ResNet, BottleneckBlock
