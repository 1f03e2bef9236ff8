"""TEST INFRASTRUCTURE ONLY -- import shim that lets the *unmodified* reference run in the build container.

This file is never imported by the product package (``breaching_amd``).  It exists so that
``oracle/make_golden.py`` can execute the real reference attacker
(``/root/reference/breaching/attacks/optimization_based_attack.py:63-88``) on CPU and write golden
fixtures into ``tests/golden/``.  ``/root/reference`` does not exist on the GPU box, so nothing that
runs there may import this module.

Why a shim is needed: ``import breaching`` pulls in ``torchvision``, ``hydra``, ``omegaconf`` and lazily
``lpips`` / ``lmdb`` (``breaching/__init__.py:3-11``, ``cases/models/model_preparation.py:4``); none of
those is installed here and there is no network.  None of them is touched by the optimisation attack
itself, so permissive stub modules are enough.
"""

import importlib.abc
import importlib.machinery
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("BREACHING_REFERENCE", "/root/reference")
_STUBBED = ("torchvision", "hydra", "omegaconf", "lpips", "lmdb", "kornia", "pytorch_wavelets")


class _StubModule(types.ModuleType):
    """A module whose every attribute is another stub (or an empty nn.Module subclass for Capitalised names)."""

    def __init__(self, name):
        super().__init__(name)
        self.__path__ = []  # makes it a package so sub-imports resolve through the finder

    def __getattr__(self, item):
        if item.startswith("__") and item.endswith("__"):
            raise AttributeError(item)
        full = f"{self.__name__}.{item}"
        if item.lstrip("_")[:1].isupper():
            import torch

            value = type(item, (torch.nn.Module,), {"__module__": self.__name__})
        else:
            value = sys.modules.get(full)
            if value is None:
                value = _StubModule(full)
                value.__spec__ = importlib.machinery.ModuleSpec(full, _FINDER, is_package=True)
                sys.modules[full] = value
        setattr(self, item, value)
        return value

    def __call__(self, *args, **kwargs):  # e.g. @hydra.main(...) used as a decorator factory
        def _decorator(fn):
            return fn

        return _decorator


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        if fullname.split(".")[0] in _STUBBED:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        return _StubModule(spec.name)

    def exec_module(self, module):
        return None


_FINDER = _StubFinder()


def have_reference():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "breaching"))


def import_reference(preload_transformers=True):
    """Return the reference ``breaching`` package, importing it through the stub finder."""
    if not have_reference():
        raise RuntimeError(f"reference tree not found at {REFERENCE_ROOT}; the shim only works in the build container")
    sys.dont_write_bytecode = True  # never write __pycache__ into the read-only reference tree
    if "breaching" in sys.modules:
        return sys.modules["breaching"]
    if preload_transformers:
        # transformers' lazy model imports crash against a stubbed torchvision (InterpolationMode.NEAREST_EXACT), so
        # resolve the model classes the tests use before the stub finder goes in.
        try:
            import transformers  # noqa: F401
            from transformers import BertConfig, BertForMaskedLM  # noqa: F401
        except ImportError:
            pass
    if _FINDER not in sys.meta_path:
        sys.meta_path.append(_FINDER)  # appended: real packages always win
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import breaching  # noqa: E402

    return breaching
