"""``passl`` — the import name the reference's launch scripts and configs use
(tools_v110/train.py:22-25: ``from passl.utils.options import parse_args`` ...; v1.1.0 shipped
``passl_v110/`` under this name).

A pure alias of ``passl_amd``: ``import passl.<x>`` returns the SAME module object as
``import passl_amd.<x>`` (one set of registries, one set of kernels), for every module at any depth:
``passl.utils / modeling / solver / hooks / datasets / engine`` (v110 surface), ``passl.models``,
``passl.engine.Engine``, ``passl.engine.loops`` (v2 surface), ``passl.loss.{moco, nt_xent, mae}`` (the
fused losses' homes) and ``passl.core`` (``grad_sync`` / ``param_sync``).
"""
import importlib
import importlib.abc
import importlib.machinery
import importlib.util
import sys

import passl_amd as _real

_PREFIX, _REAL = __name__ + '.', _real.__name__ + '.'


class _AliasFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        if not fullname.startswith(_PREFIX):
            return None
        real = _REAL + fullname[len(_PREFIX):]
        try:
            real_spec = importlib.util.find_spec(real)
        except (ImportError, AttributeError, ValueError):
            return None
        if real_spec is None:
            return None
        return importlib.machinery.ModuleSpec(fullname, self,
                                              is_package=real_spec.submodule_search_locations is not None)

    def create_module(self, spec):
        return importlib.import_module(_REAL + spec.name[len(_PREFIX):])     # the real module object itself

    def exec_module(self, module):
        pass


if not any(isinstance(f, _AliasFinder) for f in sys.meta_path):
    sys.meta_path.insert(0, _AliasFinder())

__version__ = getattr(_real, '__version__', '0.0')
