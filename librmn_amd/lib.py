"""Locates, (re)builds and loads librmn_ez_hip.so.  In-tree only: the .so lives next to this file."""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "librmn_ez_hip.so")
_lib = None


def library_path():
    return _SO


def build_library(force=False):
    """hipcc --offload-arch=gfx950 build of csrc/ (cross-compiles without a GPU)."""
    cmd = ["make", "-s", "-C", os.path.join(_HERE, "csrc")]
    if force:
        subprocess.check_call(cmd + ["clean"])
    subprocess.check_call(cmd)
    return _SO


def load_library():
    """Returns the ctypes handle of librmn_ez_hip.so; raises (loudly) when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            raise OSError(f"{_SO} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(there is no CPU fallback for the MI355X hot path)")
        _lib = ctypes.CDLL(_SO)
    return _lib
