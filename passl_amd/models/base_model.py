"""``Model`` — reference passl/models/base_model.py:25-40: a Layer with ``load_pretrained`` / ``save``."""
from ..hip import nn as hnn


class Model(hnn.Layer):
    def load_pretrained(self, path, rank=0, finetune=False):
        raise Exception('NotImplementedError, you must overwrite load_pretrained method in subclass.')

    def save(self, path, local_rank=0, rank=0):
        raise Exception('NotImplementedError, you must overwrite save method in subclass.')
