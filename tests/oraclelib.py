"""ctypes binding of oracle/liboracle.so (the plain-C CPU restatement).  TEST INFRASTRUCTURE ONLY:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the product."""
import ctypes, os, subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(_HERE, "..", "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "liboracle.so")
_lib = None

c_float_p = ctypes.POINTER(ctypes.c_float)
c_int_p = ctypes.POINTER(ctypes.c_int)


class OrcGrid(ctypes.Structure):
    _fields_ = [("grtyp", ctypes.c_char), ("grref", ctypes.c_char),
                ("ni", ctypes.c_int), ("nj", ctypes.c_int),
                ("ig", ctypes.c_int * 4), ("igref", ctypes.c_int * 4),
                ("xg", ctypes.c_float * 4), ("xgref", ctypes.c_float * 4),
                ("i1", ctypes.c_int), ("i2", ctypes.c_int), ("j1", ctypes.c_int), ("j2", ctypes.c_int),
                ("extension", ctypes.c_int),
                ("ax", c_float_p), ("ay", c_float_p), ("ncx", c_float_p), ("ncy", c_float_p),
                ("lat", c_float_p), ("lon", c_float_p)]


class OrcOpts(ctypes.Structure):
    _fields_ = [("degre_interp", ctypes.c_int), ("degre_extrap", ctypes.c_int),
                ("polar_correction", ctypes.c_int), ("vecteur", ctypes.c_int),
                ("valeur_extrap", ctypes.c_float)]


class OrcZone(ctypes.Structure):
    _fields_ = [("npts", ctypes.c_int), ("x", c_float_p), ("y", c_float_p), ("idx", c_int_p)]


class OrcGridset(ctypes.Structure):
    _fields_ = [("gdin", ctypes.POINTER(OrcGrid)), ("gdout", ctypes.POINTER(OrcGrid)),
                ("x", c_float_p), ("y", c_float_p), ("have_xy", ctypes.c_int), ("have_zones", ctypes.c_int),
                ("zones", OrcZone * 5)]


def build_oracle():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "liboracle.so"])


def oracle():
    global _lib
    if _lib is None:
        if not os.path.exists(ORACLE_SO):
            build_oracle()
        L = ctypes.CDLL(os.path.abspath(ORACLE_SO))
        L.orc_grid_define.restype = ctypes.POINTER(OrcGrid)
        L.orc_grid_define.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_char, ctypes.c_char,
                                      ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_float_p, c_float_p]
        L.orc_defset.restype = ctypes.POINTER(OrcGridset)
        L.orc_defset.argtypes = [ctypes.POINTER(OrcGrid), ctypes.POINTER(OrcGrid)]
        L.orc_supergrid_define.restype = ctypes.c_void_p
        L.orc_supergrid_define.argtypes = [ctypes.POINTER(OrcGrid), ctypes.POINTER(OrcGrid)]
        L.orc_ezyysint.argtypes = [ctypes.c_void_p, ctypes.POINTER(OrcGrid), ctypes.c_void_p, c_float_p, c_float_p]
        L.orc_ezyyuvint.argtypes = [ctypes.c_void_p, ctypes.POINTER(OrcGrid), ctypes.c_void_p, c_float_p, c_float_p, c_float_p, c_float_p]
        L.orc_ezyywdint.argtypes = L.orc_ezyyuvint.argtypes
        L.orc_cigaxg.argtypes = [ctypes.c_char] + [c_float_p] * 4 + [ctypes.c_int] * 4
        L.orc_cxgaig.argtypes = [ctypes.c_char] + [c_int_p] * 4 + [ctypes.c_float] * 4
        _lib = L
    return _lib


def fptr(a):
    if a is None:
        return ctypes.cast(None, c_float_p)
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(c_float_p)


def default_opts(**kw):
    o = OrcOpts()
    oracle().orc_default_opts(ctypes.byref(o))
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def grid_define(ni, nj, grtyp, ig=(0, 0, 0, 0), grref=" ", ax=None, ay=None):
    g = oracle().orc_grid_define(ni, nj, grtyp.encode(), grref.encode(), ig[0], ig[1], ig[2], ig[3], fptr(ax), fptr(ay))
    if not g:
        raise ValueError("orc_grid_define failed")
    return g


def np_from(ptr, n, dtype=np.float32):
    return np.ctypeslib.as_array(ptr, shape=(n,)).copy() if n else np.zeros(0, dtype)


def cxgaig(grtyp, xg1, xg2, xg3, xg4):
    ig = [ctypes.c_int() for _ in range(4)]
    oracle().orc_cxgaig(grtyp.encode(), *[ctypes.byref(i) for i in ig], xg1, xg2, xg3, xg4)
    return tuple(i.value for i in ig)
