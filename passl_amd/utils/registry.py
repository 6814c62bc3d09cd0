"""Name -> class registries and ``build_from_config``.

Mirrors the contract of the reference's passl_v110/utils/registry.py:25-133: ``register()`` as
decorator or call (keyed on ``__name__`` unless ``name=`` is given, duplicate names rejected),
``get()`` raising KeyError, and ``build_from_config(cfg, registry, default_args)`` which pops
``name`` and instantiates ``cls(**rest)``, printing and re-raising constructor errors.
"""
import inspect
import traceback


class Registry(object):
    def __init__(self, name):
        self._name = name
        self._obj_map = {}

    def _do_register(self, name, obj):
        assert name not in self._obj_map, \
            "An object named '{}' was already registered in '{}' registry!".format(name, self._name)
        self._obj_map[name] = obj

    def register(self, obj=None, name=None):
        if obj is None:
            def deco(func_or_class, name=name):
                self._do_register(name or func_or_class.__name__, func_or_class)
                return func_or_class
            return deco
        self._do_register(name or obj.__name__, obj)

    def get(self, name):
        ret = self._obj_map.get(name)
        if ret is None:
            raise KeyError("No object named '{}' found in '{}' registry!".format(name, self._name))
        return ret

    def __contains__(self, name):
        return name in self._obj_map


def build_from_config(cfg, registry, default_args=None):
    if not isinstance(cfg, dict):
        raise TypeError('cfg must be a dict, but got {}'.format(type(cfg)))
    if 'name' not in cfg and (default_args is None or 'name' not in default_args):
        raise KeyError('`cfg` or `default_args` must contain the key "name", but got {}\n{}'.format(
            cfg, default_args))
    if not isinstance(registry, Registry):
        raise TypeError('registry must be an Registry object, but got {}'.format(type(registry)))
    if not (isinstance(default_args, dict) or default_args is None):
        raise TypeError('default_args must be a dict or None, but got {}'.format(type(default_args)))
    args = dict(cfg)
    if default_args is not None:
        for k, v in default_args.items():
            args.setdefault(k, v)
    cls_name = args.pop('name')
    if isinstance(cls_name, str):
        obj_cls = registry.get(cls_name)
    elif inspect.isclass(cls_name):
        obj_cls = cls_name
    else:
        raise TypeError('name must be a str or valid name, but got {}'.format(type(cls_name)))
    try:
        return obj_cls(**args)
    except Exception as e:
        print('Fail to initial class [{}] with error: {} and stack:\n{}'.format(
            cls_name, e, traceback.format_exc()))
        raise e
