"""hite_amd -- MI355X-native (gfx950 / HIP) implementation of HiTE's dynamic-boundary-adjustment
hot path behind a C ABI (include/hite_gpu.h).  See DESIGN.md / INTEGRATION.md."""
from ._lib import Context, HiteError, load, SO_PATH  # noqa: F401

__all__ = ["Context", "HiteError", "load", "SO_PATH"]
