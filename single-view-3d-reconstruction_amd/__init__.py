"""MI355X-native IF-Net occupancy-query path (drop-in for the reference's model/ifnet.py and
model/projection.py hot path).  All arithmetic runs in hand-written gfx950 HIP kernels behind
the C ABI of include/svr_hip.h; this package is the host-side mirror of the reference's
nn.Module / training_step interface.

The directory name is not a Python identifier; import it through the repo-root alias
``import svr_amd`` (or importlib.import_module("single-view-3d-reconstruction_amd")).
"""
import os

# The training step runs on up to three HIP streams (main + two side streams, model/ifnet.py) beside RCCL's own; the HIP
# runtime multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4), and with a process group alive two of
# ours shared a queue: 19.8 instead of 18.4 ms/step.  Read when the runtime initialises (first HIP call), so setting it
# at import is early enough; an explicit setting of the user wins.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from . import _lib  # noqa: F401,E402

__all__ = ["_lib"]
