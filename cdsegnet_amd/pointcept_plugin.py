"""Plug the MI355X models into the REFERENCE's own registry, so that ``tools/test_*.py`` runs unchanged.

The reference builds its model with ``pointcept.models.builder.MODELS.build(cfg.model)``
(ref: pointcept/models/builder.py:9-16, engines/test.py:55-59) and registers
``"DefaultSegmentorV2"`` (models/default.py:13-14) and ``"PT-v3m1"``
(models/point_transformer_v3/point_transformer_v3m1_base.py:1340) at import time.

    import pointcept.models                      # the reference registers its classes
    import cdsegnet_amd.pointcept_plugin as plug
    plug.register_into()                         # ... and ours replace the two names (force=True)

or, with NO edit on the reference side at all (every interpreter of the run, including the
``mp.spawn`` workers of pointcept/engines/launch.py:35-135, inherits the environment):

    PYTHONPATH=<repo>/cdsegnet_amd/site:<repo> python tools/test_CDSegNet_ScanNet.py

``cdsegnet_amd/site/sitecustomize.py`` calls :func:`install_import_hook`, which re-registers the two
names right after ``pointcept.models`` has been imported.
"""
import importlib
import importlib.abc
import importlib.util
import sys

NAMES = ("DefaultSegmentorV2", "PT-v3m1")


def register_into(registry=None, force=True):
    """Register the MI355X ``DefaultSegmentorV2`` / ``PT-v3m1`` into `registry` - any object with the reference
    Registry's ``register_module(name=None, force=False, module=None)`` (ref: pointcept/utils/registry.py:266-316);
    default: the reference's ``pointcept.models.builder.MODELS``.  Returns the registry."""
    from . import models  # noqa: F401  (fills cdsegnet_amd.registry.MODELS)
    from .registry import MODELS as OURS
    if registry is None:
        registry = importlib.import_module("pointcept.models.builder").MODELS
    for name in NAMES:
        cls = OURS.get(name)
        if cls is None:
            raise KeyError(f"{name} is not registered in cdsegnet_amd")
        registry.register_module(name=name, force=force, module=cls)
    return registry


class _PostImport(importlib.abc.MetaPathFinder):
    """Runs `callback` once, right after module `target` has been executed."""

    def __init__(self, target, callback):
        self.target, self.callback, self.busy = target, callback, False

    def find_spec(self, fullname, path=None, target=None):
        if fullname != self.target or self.busy:
            return None
        self.busy = True
        try:
            spec = importlib.util.find_spec(fullname)
        finally:
            self.busy = False
        if spec is None or spec.loader is None:
            return None
        loader, callback, finder = spec.loader, self.callback, self
        orig_exec = loader.exec_module

        def exec_module(module):
            orig_exec(module)
            if finder in sys.meta_path:
                sys.meta_path.remove(finder)
            callback()

        loader.exec_module = exec_module
        return spec


def install_import_hook(target="pointcept.models"):
    """Arrange for :func:`register_into` to run as soon as the reference's model package has been imported."""
    if target in sys.modules:
        register_into()
        return None
    hook = _PostImport(target, register_into)
    sys.meta_path.insert(0, hook)
    return hook
