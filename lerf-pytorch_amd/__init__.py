"""lerf-pytorch_amd: the LeRF LUT resampling hot path on MI355X (gfx950).

Importable as `lerf_pytorch_amd` (see lerf_pytorch_amd.py at the repo root).
Host side = Python mirroring the reference's class / function API; device side
= hand-written HIP behind a C ABI (include/lerf_hip.h, liblerf_hip.so).
"""
from . import _lib  # noqa: F401
from ._lib import LerfError  # noqa: F401
from .luts import LutSet, load_lut_arrays  # noqa: F401
from .pipeline import LerfEngine, sr, warp  # noqa: F401
from . import metrics, stream  # noqa: F401

__all__ = ["LerfEngine", "LutSet", "load_lut_arrays", "sr", "warp", "LerfError", "LIB_PATH"]


def __getattr__(name):
    # LIB_PATH follows _lib.use_library() (tools load variant builds): looked up, not copied at import
    if name == "LIB_PATH":
        return _lib.LIB_PATH
    raise AttributeError("module %r has no attribute %r" % (__name__, name))
