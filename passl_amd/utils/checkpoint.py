"""Checkpoint / weight-exchange helpers.

File format = a pickled dict of numpy arrays, the layout of the reference's ``save`` helper
(passl_v110/hooks/checkpoint_hook.py:23-50) and of ``paddle.save`` for a dygraph ``state_dict``
(which adds the bookkeeping key ``StructuredToParameterName@@``  [Paddle-semantics]).  Because
``state_dict()`` here uses the reference's key names and logical shapes, a published ``.pdparams``
loads without any transposition."""
import pickle

import numpy as np
import torch

PADDLE_META_KEYS = ('StructuredToParameterName@@',)


def load_pickle(path):
    with open(path, 'rb') as f:
        obj = pickle.load(f, encoding='latin1')
    if isinstance(obj, dict):
        for k in PADDLE_META_KEYS:
            obj.pop(k, None)
    return obj


def to_numpy(obj):
    if torch.is_tensor(obj):
        t = obj.detach().cpu()
        return t.float().contiguous().numpy() if t.is_floating_point() else t.numpy()
    if isinstance(obj, dict):
        return {k: to_numpy(v) for k, v in obj.items()}
    return obj


def to_tensors(sd):
    return {k: torch.as_tensor(np.asarray(v)) if isinstance(v, (np.ndarray, np.generic)) else v
            for k, v in sd.items()}


def load_state_into(model, state_dict, strict=False):
    """state_dict: name -> numpy array (or tensor).  A bare backbone dict (as written by
    tools/extract_weight.py --remove_prefix) can be loaded into ``model.backbone``."""
    sd = to_tensors(state_dict)
    return model.load_state_dict(sd, strict=strict)


def load_lenient(module, state_dict, logger=None, what='weights'):
    """Paddle's ``set_state_dict`` policy: load what matches, WARN per missing / unexpected key, and
    refuse a file in which nothing matched (a wrong key prefix would otherwise "load" successfully and
    leave the model on its random init)."""
    import logging
    logger = logger or logging.getLogger('passl')
    sd = to_tensors(state_dict)
    own = module.state_dict()
    matched = [k for k in sd if k in own]
    if own and not matched:
        raise ValueError('%s: none of the %d keys matches the model (first file key %r, first model '
                         'key %r) - wrong prefix?' % (what, len(sd), next(iter(sd), None),
                                                      next(iter(own), None)))
    res = module.load_state_dict(sd, strict=False)
    for k in res.missing_keys:
        logger.warning('%s: %s is not found in the provided dict.' % (what, k))
    for k in res.unexpected_keys:
        logger.warning('%s: skip loading for %s (not in the model).' % (what, k))
    return res
