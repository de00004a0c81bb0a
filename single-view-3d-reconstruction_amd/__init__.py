"""MI355X-native IF-Net occupancy-query path (drop-in for the reference's model/ifnet.py and
model/projection.py hot path).  All arithmetic runs in hand-written gfx950 HIP kernels behind
the C ABI of include/svr_hip.h; this package is the host-side mirror of the reference's
nn.Module / training_step interface.

The directory name is not a Python identifier; import it through the repo-root alias
``import svr_amd`` (or importlib.import_module("single-view-3d-reconstruction_amd")).
"""
from . import _lib  # noqa: F401

__all__ = ["_lib"]
