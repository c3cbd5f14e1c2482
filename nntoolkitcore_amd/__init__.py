"""nntoolkitcore_amd -- MI355X-native (gfx950) implementation of NNToolkitCore's
time-series inference hot path behind the reference's own C layer API.

The product is the shared library ``lib/libnntoolkitcore_hip.so`` (C host layer +
hand-written HIP kernels; boundary: ``include/nntoolkitcore_hip.h``).  This package
only holds its sources (``csrc/``), the build script and a ctypes binding used by
the tests and the benchmark.
"""
from . import capi  # noqa: F401

__all__ = ["capi", "layers"]
