"""damar_amd -- MI355X-native daligner overlap hot path (DAmar / MARVEL).

The product is native code: ``libdamar_hip.so`` (hand-written HIP kernels for gfx950
behind the reference's ``dalign/filter.h`` C interface) and the ``daligner`` host binary
(C).  This package only holds the ctypes mirror of that C interface used by the tests and
``bench.py``; it contains no compute and no CPU fallback.
"""
from .lib import lib_path, load, LibraryMissing          # noqa: F401
from . import api                                         # noqa: F401

__all__ = ["api", "lib_path", "load", "LibraryMissing"]
