"""`sample_factory` import surface for the MI355X-native engine.

A training script written against Sample Factory —

    from sample_factory.cfg.arguments import parse_full_cfg, parse_sf_args
    from sample_factory.envs.env_utils import register_env
    from sample_factory.algo.utils.context import global_model_factory
    from sample_factory.train import run_rl

— runs unchanged on this engine when this repository is first on `sys.path`: every `sample_factory.<module>` resolves
to `sample_factory_amd.<module>` (the same module object under both names, so registries are shared).  Modules of the
reference that are outside the hot-path scope do not exist and fail with ModuleNotFoundError.
"""
import importlib
import importlib.abc
import importlib.util
import sys

import sample_factory_amd as _impl

_PREFIX, _REAL = "sample_factory.", "sample_factory_amd."


class _AliasFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        if not fullname.startswith(_PREFIX):
            return None
        real = _REAL + fullname[len(_PREFIX):]
        try:
            real_spec = importlib.util.find_spec(real)
            if real_spec is None:
                return None
        except ModuleNotFoundError:
            return None
        return importlib.util.spec_from_loader(fullname, self, origin=real_spec.origin,
                                               is_package=hasattr(importlib.import_module(real), "__path__"))

    def create_module(self, spec):
        real = importlib.import_module(_REAL + spec.name[len(_PREFIX):])
        # the import machinery is about to stamp the ALIAS spec (and loader / package) onto whatever create_module
        # returns; the object is the real module, whose own spec must survive (importlib.reload, spec-based tooling,
        # __package__ == __spec__.parent): remember it here, put it back in exec_module
        self._real_attrs[id(real)] = {k: getattr(real, k) for k in ("__spec__", "__loader__", "__package__", "__name__",
                                                                      "__path__", "__file__", "__cached__")
                                      if hasattr(real, k)}
        return real

    def exec_module(self, module):  # the real module is already initialised: only undo the alias stamping
        for k, v in self._real_attrs.pop(id(module), {}).items():
            setattr(module, k, v)

    _real_attrs: dict = {}

    # `python -m sample_factory.<module>` (runpy) asks the loader for the code object: hand over the real module's
    def _real_loader(self, fullname):
        real = _REAL + fullname[len(_PREFIX):]
        return real, importlib.util.find_spec(real).loader

    def get_code(self, fullname):
        real, loader = self._real_loader(fullname)
        return loader.get_code(real)

    def get_source(self, fullname):
        real, loader = self._real_loader(fullname)
        return loader.get_source(real)

    def is_package(self, fullname):
        real, loader = self._real_loader(fullname)
        return loader.is_package(real)


if not any(isinstance(f, _AliasFinder) for f in sys.meta_path):
    sys.meta_path.insert(0, _AliasFinder())

__version__ = getattr(_impl, "__version__", "0")
