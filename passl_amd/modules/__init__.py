from . import init, freeze
from .freeze import freeze_batchnorm_statictis
