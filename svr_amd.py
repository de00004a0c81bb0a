"""Import alias: ``import svr_amd`` == the package directory ``single-view-3d-reconstruction_amd``
(whose name is not a valid Python identifier)."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("single-view-3d-reconstruction_amd")
sys.modules[__name__] = _pkg
