from .builder import HEADS, build_head
from .contrastive_head import ContrastiveHead
from .simclr_contrastive_head import SimCLRContrastiveHead
from .clip_head import CLIPHead
from .clas_head import ClasHead, VisionTransformerClsHead
