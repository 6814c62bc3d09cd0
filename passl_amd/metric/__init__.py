"""v2 metric front end — reference passl/metric/__init__.py:22-49 (``build_metrics`` -> ``CombinedMetrics``: a yaml
list of ``{Name: kwargs}`` entries merged into one ordered dict) and passl/metric/metrics.py:29-56 (``TopkAcc``)."""
import copy
from collections import OrderedDict

from .metrics import TopkAcc

_METRICS = {'TopkAcc': TopkAcc}


class CombinedMetrics(object):
    def __init__(self, config_list):
        self.metric_func_list = []
        assert isinstance(config_list, list), 'operator config should be a list'
        for config in config_list:
            assert isinstance(config, dict) and len(config) == 1, 'yaml format error'
            name = list(config)[0]
            params = config[name]
            if name not in _METRICS:
                raise NotImplementedError('metric %r is not on the linear-probe path (built: %s)'
                                          % (name, sorted(_METRICS)))
            self.metric_func_list.append(_METRICS[name](**params) if params is not None else _METRICS[name]())

    def __call__(self, *args, **kwargs):
        metric_dict = OrderedDict()
        for metric_func in self.metric_func_list:
            metric_dict.update(metric_func(*args, **kwargs))
        return metric_dict


def build_metrics(config):
    return CombinedMetrics(copy.deepcopy(config))
