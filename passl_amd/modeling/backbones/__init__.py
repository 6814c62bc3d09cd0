from .builder import BACKBONES, build_backbone
from .resnet import ResNet, BottleneckBlock
from .resnetsimclr import ResNetsimclr, ResNetCifar
from .mae import MAE, MAE_ViT
from .vision_transformer import VisionTransformer
from .clip import CLIP
