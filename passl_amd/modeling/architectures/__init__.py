from .builder import MODELS, build_model
from .moco import MoCo
from .simclr import SimCLR
from .MAE import MAE_PRETRAIN, MAE_FINETUNE
from .CLIPWrapper import CLIPWrapper
from .clas import Classification
