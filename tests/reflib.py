"""Loader for oracle/_ref/libezref.so (the reference's own interp sources, built by
oracle/build_ref.sh).  TEST INFRASTRUCTURE ONLY.

The library keeps a few FST file-I/O symbols undefined (grid-from-file code that the hot
path never reaches), so it must be opened with lazy binding; ctypes forces RTLD_NOW, hence
the explicit dlopen through libc.
"""
import ctypes, os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
REF_SO = os.path.join(_HERE, "..", "oracle", "_ref", "libezref.so")
_lib = None

def have_ref():
    return os.path.exists(REF_SO)

class _NoReference:
    """FUZZ_PRODUCT_ONLY=1: every reference entry point is a no-op that returns 1 and leaves its outputs alone.  A fuzz tool run this way makes exactly
    the product calls of a normal run of the same seed and none of the reference's: whether it survives tells a crash of the product from the
    reference's own heap corruption (tests/test_gpu_vs_reference_build.py); what it prints is meaningless"""
    def __getattr__(self, name):
        def nothing(*args, **kwargs):
            return 1
        return nothing


def ref():
    global _lib
    if _lib is None and os.environ.get("FUZZ_PRODUCT_ONLY"):
        _lib = _NoReference()
    if _lib is None:
        libc = ctypes.CDLL(None)
        libc.dlopen.restype = ctypes.c_void_p
        libc.dlopen.argtypes = [ctypes.c_char_p, ctypes.c_int]
        h = libc.dlopen(os.path.abspath(REF_SO).encode(), os.RTLD_LAZY | os.RTLD_LOCAL)
        if not h:
            raise OSError("cannot dlopen " + REF_SO)
        _lib = ctypes.CDLL(REF_SO, handle=h)
    return _lib

def fptr(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))

def iptr(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))
