"""YAML config -> nested ``AttrDict`` with ``-o a.b.0.c=v`` overrides.

Behavioural contract (what the reference's configs and launch lines rely on,
passl_v110/utils/config.py:25-128):
  * every mapping becomes an ``AttrDict`` (attribute access), at any depth, also inside lists;
  * string leaves are passed through ``ast.literal_eval`` when they parse (``"[0.2, 1.]"`` -> list,
    ``"1.0/255.0"`` stays a string because it is an expression, not a literal);
  * an override addresses an EXISTING key (dict) or index (list) by a dotted path; its value is
    evaluated as a Python expression when possible, kept as a string otherwise;
  * a missing file, a malformed option or an unknown key is an ``AssertionError``.
"""
import copy
import os
from ast import literal_eval

import yaml

__all__ = ['AttrDict', 'get_config', 'parse_config', 'override_config', 'create_attr_dict']


class AttrDict(dict):
    """dict whose items are also attributes."""

    def __getattr__(self, name):
        if name in self:
            return self[name]
        raise AttributeError(name)

    def __setattr__(self, name, value):
        if name in self.__dict__:
            object.__setattr__(self, name, value)
        else:
            self[name] = value

    def __deepcopy__(self, memo):
        return AttrDict((k, copy.deepcopy(v, memo)) for k, v in self.items())


def _literal(value):
    """str -> Python literal when it is one."""
    if not isinstance(value, str):
        return value
    try:
        return literal_eval(value)
    except (ValueError, SyntaxError, TypeError, MemoryError, RecursionError):
        return value


def _convert(node):
    """Recursively wrap mappings and evaluate string literals (lists are converted in place)."""
    if isinstance(node, dict):
        wrapped = node if isinstance(node, AttrDict) else AttrDict(node)
        for key in list(wrapped):
            wrapped[key] = _convert(wrapped[key])
        return wrapped
    if isinstance(node, list):
        for i, item in enumerate(node):
            node[i] = _convert(item) if isinstance(item, (dict, list)) else item
        return node
    return _literal(node)


def create_attr_dict(cfg):
    """In-place conversion of a (possibly plain) mapping tree."""
    for key in list(cfg):
        cfg[key] = _convert(cfg[key])


def parse_config(cfg_file):
    with open(cfg_file, 'r') as f:
        return _convert(yaml.load(f, Loader=yaml.SafeLoader))


def _evaluate(text):
    try:
        return eval(text)            # the reference evaluates override values the same way
    except Exception:
        return text


def _assign(node, path, text):
    """Walk ``path`` (list of keys / indices) and replace the addressed leaf."""
    assert isinstance(node, (list, dict)), '{} should be a list or a dict'.format(node)
    assert path, 'empty override key'
    head = _evaluate(path[0]) if isinstance(node, list) else path[0]
    if isinstance(node, list):
        assert isinstance(head, int) and head < len(node), 'index({}) out of range({})'.format(head, node)
    else:
        assert head in node, '{} is not exist in {}'.format(head, node)
    if len(path) == 1:
        node[head] = _evaluate(text)
    else:
        _assign(node[head], path[1:], text)


def override_config(config, options=None):
    for opt in options or ():
        assert isinstance(opt, str), 'option({}) should be a str'.format(opt)
        assert opt.count('=') == 1, 'option({}) must be key=value with a single ='.format(opt)
        dotted, text = opt.split('=')
        _assign(config, dotted.split('.'), text)
    return config


def get_config(fname, overrides=None):
    assert os.path.exists(fname), 'config file({}) is not exist'.format(fname)
    return override_config(parse_config(fname), overrides)
