"""Import shim for the upstream reference (fixture generation ONLY, this container only).

The reference lives read-only at /root/reference and cannot travel to the GPU box.
This module makes `import ccsd...` work here by providing inert stand-ins for the
third-party packages the reference imports at module top but never touches on the
predictor-corrector sampling path (rdkit, toponetx, easydict, wandb, ...).  It is
used only by tools/make_golden.py to capture golden input/output vectors that are
committed under tests/golden/.  Nothing in ccsd_amd/, tests/, bench.py or
__graft_entry__.py imports this file.
"""
from __future__ import annotations

import importlib.abc
import importlib.machinery
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("CCSD_REFERENCE_ROOT", "/root/reference")

_STUB_ROOTS = (
    "rdkit", "toponetx", "pyemd", "wandb", "moses", "imageio", "hypernetx",
    "freezegun", "kaleido", "molsets", "fcd_torch",
)


class _Inert(types.ModuleType):
    """Module whose every attribute is another inert object (callable, subscriptable)."""

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        obj = _InertObj(f"{self.__name__}.{name}")
        setattr(self, name, obj)
        return obj


class _InertObj:
    def __init__(self, name="inert"):
        self._name = name

    def __call__(self, *a, **k):
        return _InertObj(self._name + "()")

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _InertObj(self._name + "." + name)

    def __getitem__(self, k):
        return _InertObj(self._name + "[]")

    def __iter__(self):
        return iter(())

    def __mro_entries__(self, bases):
        return (object,)


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        root = fullname.split(".")[0]
        if root in _STUB_ROOTS:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _Inert(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        pass


class EasyDict(dict):
    """Minimal functional EasyDict: dict with recursive attribute access."""

    def __init__(self, d=None, **kw):
        super().__init__()
        d = dict(d or {})
        d.update(kw)
        for k, v in d.items():
            self[k] = v

    @classmethod
    def _wrap(cls, v):
        if isinstance(v, dict) and not isinstance(v, EasyDict):
            return cls(v)
        if isinstance(v, (list, tuple)):
            return type(v)(cls._wrap(x) for x in v)
        return v

    def __setitem__(self, k, v):
        super().__setitem__(k, self._wrap(v))

    def __setattr__(self, k, v):
        self[k] = v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def update(self, *a, **k):
        for kk, v in dict(*a, **k).items():
            self[kk] = v

    # pickled EasyDicts restore through __setstate__/__dict__ updates
    def __setstate__(self, state):
        for k, v in state.items():
            self[k] = v


def install():
    """Install the stubs and put the reference on sys.path. Idempotent."""
    if getattr(install, "_done", False):
        return
    sys.dont_write_bytecode = True
    if not os.path.isdir(REFERENCE_ROOT):
        raise RuntimeError(f"reference not present at {REFERENCE_ROOT}")
    sys.meta_path.insert(0, _StubFinder())
    ed = types.ModuleType("easydict")
    ed.EasyDict = EasyDict
    sys.modules["easydict"] = ed
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    install._done = True


def load_reference_ckpt(relpath):
    """torch.load a reference checkpoint (weights_only=False: it pickles EasyDict)."""
    import torch

    install()
    return torch.load(os.path.join(REFERENCE_ROOT, relpath), map_location="cpu", weights_only=False)
