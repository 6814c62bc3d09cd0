from .sync_utils import GradReducer, grad_sync, param_sync
