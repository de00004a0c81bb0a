"""Import alias: ``import svr_amd`` == the package directory ``single-view-3d-reconstruction_amd``
(whose name is not a valid Python identifier).  Submodules are aliased too: ``svr_amd.model.ifnet`` IS
``single-view-3d-reconstruction_amd.model.ifnet`` (one module object, not a second copy of it)."""
import importlib
import importlib.abc
import importlib.util
import os
import sys

_REAL = "single-view-3d-reconstruction_amd"
_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)


class _AliasLoader(importlib.abc.Loader):
    def __init__(self, real):
        self.real = real

    def create_module(self, spec):
        return importlib.import_module(self.real)

    def exec_module(self, module):          # already executed under its real name
        pass


class _AliasFinder(importlib.abc.MetaPathFinder):
    def find_spec(self, fullname, path=None, target=None):
        if not fullname.startswith(__name__ + "."):
            return None
        real = _REAL + fullname[len(__name__):]
        if importlib.util.find_spec(real) is None:
            return None
        return importlib.util.spec_from_loader(fullname, _AliasLoader(real))


if not any(isinstance(f, _AliasFinder) for f in sys.meta_path):
    sys.meta_path.insert(0, _AliasFinder())
_pkg = importlib.import_module(_REAL)
sys.modules[__name__] = _pkg
