"""freeze_batchnorm_statictis (sic): make every BatchNorm of ``layer`` normalise with its running
statistics and stop updating them — reference passl_v110/modules/freeze.py:18-23."""
from ..hip.nn import _BatchNormBase


def freeze_batchnorm_statictis(layer):
    def freeze_bn(m):
        if isinstance(m, _BatchNormBase):
            m._use_global_stats = True

    layer.apply(freeze_bn)
