from .builder import NECKS, build_neck
from .base_neck import LinearNeck, NonLinearNeckV1, NonLinearNeckfc3
