from .misc import AverageMeter
from .registry import Registry, build_from_config
from .config import AttrDict, get_config
