"""Name -> class registries and ``build_from_config``.

Contract (the one the reference's builders and YAML files rely on, passl_v110/utils/registry.py:25-133):
``register`` works as ``@R.register()``, ``@R.register(name=...)`` or ``R.register(obj, name=...)`` and keys
on ``__name__`` by default (a duplicate key is an ``AssertionError``); ``get`` raises ``KeyError``;
``build_from_config(cfg, registry, default_args)`` merges the defaults under ``cfg``, pops ``name`` (a
registered key or a class) and instantiates it with the remaining items, printing the traceback of a
failing constructor before re-raising.
"""
import inspect
import traceback


class Registry(object):
    def __init__(self, name):
        self._name = name
        self._obj_map = {}

    def __contains__(self, key):
        return key in self._obj_map

    def __len__(self):
        return len(self._obj_map)

    def _add(self, obj, key=None):
        key = key or obj.__name__
        assert key not in self._obj_map, \
            "An object named '{}' was already registered in '{}' registry!".format(key, self._name)
        self._obj_map[key] = obj
        return obj

    def register(self, obj=None, name=None):
        if obj is not None:                       # plain call: R.register(cls) / R.register(cls, name='x')
            self._add(obj, name)
            return None
        return lambda target: self._add(target, name)     # decorator form

    def get(self, name):
        try:
            return self._obj_map[name]
        except KeyError:
            raise KeyError("No object named '{}' found in '{}' registry!".format(name, self._name)) from None


def _resolve(target, registry):
    if isinstance(target, str):
        return registry.get(target)
    if inspect.isclass(target):
        return target
    raise TypeError('name must be a str or valid name, but got {}'.format(type(target)))


def build_from_config(cfg, registry, default_args=None):
    for what, value, ok in (('cfg', cfg, isinstance(cfg, dict)),
                            ('registry', registry, isinstance(registry, Registry)),
                            ('default_args', default_args, default_args is None or isinstance(default_args, dict))):
        if not ok:
            raise TypeError('{} has the wrong type: {}'.format(what, type(value)))
    kwargs = dict(default_args or {})
    kwargs.update(cfg)
    if 'name' not in kwargs:
        raise KeyError('`cfg` or `default_args` must contain the key "name", but got {}\n{}'.format(
            cfg, default_args))
    target = kwargs.pop('name')
    obj_cls = _resolve(target, registry)
    try:
        return obj_cls(**kwargs)
    except Exception as err:
        print('Fail to initial class [{}] with error: {} and stack:\n{}'.format(target, err, traceback.format_exc()))
        raise
