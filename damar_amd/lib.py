"""Locate and load libdamar_hip.so (built in-tree by damar_amd/csrc/Makefile)."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class LibraryMissing(RuntimeError):
    """The HIP extension is not built / not loadable.  There is no fallback."""


def lib_path():
    return os.path.join(_HERE, "libdamar_hip.so")


def bin_path(name):
    return os.path.join(_HERE, "bin", name)


def load():
    """Return the loaded C-ABI library or raise LibraryMissing (never a CPU stand-in)."""
    global _LIB
    if _LIB is None:
        p = lib_path()
        if not os.path.exists(p):
            raise LibraryMissing(
                "%s not found: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or make -C damar_amd/csrc); damar_amd has no CPU fallback" % p)
        try:
            _LIB = ctypes.CDLL(p)
        except OSError as e:            # pragma: no cover - depends on the machine
            raise LibraryMissing("cannot load %s: %s" % (p, e))
    return _LIB
