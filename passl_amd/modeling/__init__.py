from .backbones import build_backbone, BACKBONES
from .necks import build_neck, NECKS
from .heads import build_head, HEADS
from .architectures import build_model, MODELS
