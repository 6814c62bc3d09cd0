from .hook import Hook
from .builder import HOOKS, build_hook
from .optimizer_hook import OptimizerHook
from .timer_hook import IterTimerHook
from .log_hook import LogHook
from .lr_scheduler_hook import LRSchedulerHook
from .checkpoint_hook import CheckpointHook
from .evaluate_hook import EvaluateHook
