"""YAML -> nested AttrDict with ``-o a.b.0.c=v`` overrides.

Same behaviour as the reference's passl_v110/utils/config.py:25-128: string leaves go through
``literal_eval`` (so ``1.0/255.0`` stays a string but ``[0.2, 1.]`` becomes a list), overrides
are ``eval``-ed when possible, must address an existing key / index, and ``get_config`` asserts
that the file exists.
"""
import os
from ast import literal_eval

import yaml

__all__ = ['AttrDict', 'get_config', 'parse_config', 'override_config', 'create_attr_dict']


class AttrDict(dict):
    def __getattr__(self, key):
        try:
            return self[key]
        except KeyError:
            raise AttributeError(key)

    def __setattr__(self, key, value):
        if key in self.__dict__:
            self.__dict__[key] = value
        else:
            self[key] = value

    def __deepcopy__(self, memo):
        import copy
        return AttrDict({k: copy.deepcopy(v, memo) for k, v in self.items()})


def create_attr_dict(cfg):
    for key, value in list(cfg.items()):
        if type(value) is dict:
            cfg[key] = value = AttrDict(value)
        if isinstance(value, str):
            try:
                value = literal_eval(value)
            except BaseException:
                pass
        if isinstance(value, AttrDict):
            create_attr_dict(cfg[key])
        elif isinstance(value, list):
            _recurse_list(value)
        else:
            cfg[key] = value
    return None


def _recurse_list(lst):
    for i, v in enumerate(lst):
        if type(v) is dict or isinstance(v, AttrDict):
            lst[i] = v if isinstance(v, AttrDict) else AttrDict(v)
            create_attr_dict(lst[i])
        elif isinstance(v, list):
            _recurse_list(v)


def parse_config(cfg_file):
    with open(cfg_file, 'r') as f:
        cfg = AttrDict(yaml.load(f, Loader=yaml.SafeLoader))
    create_attr_dict(cfg)
    return cfg


def override(dl, ks, v):
    def str2num(s):
        try:
            return eval(s)
        except Exception:
            return s

    assert isinstance(dl, (list, dict)), '{} should be a list or a dict'.format(dl)
    assert len(ks) > 0, 'lenght of keys should larger than 0'
    if isinstance(dl, list):
        k = str2num(ks[0])
        if len(ks) == 1:
            assert k < len(dl), 'index({}) out of range({})'.format(k, dl)
            dl[k] = str2num(v)
        else:
            override(dl[k], ks[1:], v)
    else:
        if len(ks) == 1:
            assert ks[0] in dl, '{} is not exist in {}'.format(ks[0], dl)
            dl[ks[0]] = str2num(v)
        else:
            override(dl[ks[0]], ks[1:], v)


def override_config(config, options=None):
    if options is not None:
        for opt in options:
            assert isinstance(opt, str), 'option({}) should be a str'.format(opt)
            assert '=' in opt, 'option({}) should contain a = to distinguish between key and value'.format(opt)
            pair = opt.split('=')
            assert len(pair) == 2, 'there can be only a = in the option'
            key, value = pair
            override(config, key.split('.'), value)
    return config


def get_config(fname, overrides=None):
    assert os.path.exists(fname), 'config file({}) is not exist'.format(fname)
    config = parse_config(fname)
    override_config(config, overrides)
    return config
