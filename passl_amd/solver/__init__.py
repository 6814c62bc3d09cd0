from .builder import (build_lr_scheduler, build_lr_scheduler_simclr, build_optimizer, LRSCHEDULERS,
                      OPTIMIZERS)
from . import lr_scheduler, optimizer
