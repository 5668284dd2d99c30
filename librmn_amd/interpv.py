"""Python mirror of librmn's 1-D (vertical) interpolation package, src/interpv, bound to librmn_ez_hip.so.

Host calls go through the reference's own Fortran-callable symbols (`interp1d_linear_`, `interp1d_linear8_`,
... include/interpv_hip.h): numpy arrays shaped (levels, ijDim), C order = the reference's Fortran (ijDim, levels).
Device calls (`*_dev`) take torch tensors of the same shape resident in HBM.
"""
import ctypes
import numpy as np

from .lib import load_library

NONE, NEAREST, LINEAR, CUBIC_LAGRANGE, CUBIC_DERIVS = 0, 1, 2, 3, 4
X_NONE, X_FIXED, X_LAPSERATE, X_ABORT = 0, 1, 2, 3
INTERP_SYMBOL = {NEAREST: "interp1d_nearestneighbour", LINEAR: "interp1d_linear", CUBIC_LAGRANGE: "interp1d_cubiclagrange",
                 CUBIC_DERIVS: "interp1d_cubicwithderivs"}
EXTRAP_SYMBOL = {X_FIXED: "extrap1d_fixed", X_LAPSERATE: "extrap1d_lapserate", X_ABORT: "extrap1d_abort"}
_configured = False


def _lib():
    global _configured
    L = load_library()
    if not _configured:
        vp, ci, cd = ctypes.c_void_p, ctypes.c_int, ctypes.c_double
        L.interpv_hip_findpos_dev.argtypes = [ci] * 6 + [vp, vp, vp]
        L.interpv_hip_interp_dev.argtypes = [ci] * 7 + [vp] * 7 + [ci, ci]
        L.interpv_hip_extrap_dev.argtypes = [ci] * 7 + [vp] * 7 + [ci, ci, cd, cd]
        L.interpv_hip_column_dev.argtypes = [ci] * 8 + [vp] * 7 + [ci, ci, cd, cd]
        _configured = True
    return L


def _p(a):
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        return a.ctypes.data_as(ctypes.c_void_p)
    return ctypes.c_void_p(a.data_ptr())


def _ibr(v):
    return ctypes.byref(ctypes.c_int32(int(v)))


def _prec8(a):
    import torch
    if isinstance(a, np.ndarray):
        return 1 if a.dtype == np.float64 else 0
    return 1 if a.dtype == torch.float64 else 0


# ---------------------------------------------------------------- host arrays, the reference's symbols
def findpos(n, vls, vld):
    """Interp1D_FindPos / Interp1D_FindPos8: returns posnDestInSrc (destNumLevels, dst_ijDim), int32, 1-based"""
    ns, sij = vls.shape; nd, dij = vld.shape
    posn = np.full((nd, dij), -999, np.int32)
    f = getattr(_lib(), "interp1d_findpos8_" if vls.dtype == np.float64 else "interp1d_findpos_")
    f.restype = None
    f(_ibr(n), _ibr(ns), _ibr(nd), _ibr(sij), _ibr(dij), _p(vls), _p(posn), _p(vld))
    return posn


def fortran_apply(symbol, n, vls, ss, sds, posn, vld, sd, sdd, xdown=False, xup=False, gdown=0.0, gup=0.0, extended=False):
    """<symbol>_ / <symbol>8_ (or _x_ / _x8_): works in place on sd / sdd"""
    ns, sij = vls.shape; nd, dij = vld.shape
    p8 = vls.dtype == np.float64
    R = ctypes.c_double if p8 else ctypes.c_float
    name = symbol + (("_x8_" if p8 else "_x_") if extended else ("8_" if p8 else "_"))
    f = getattr(_lib(), name)
    f.restype = None
    args = [_ibr(n), _ibr(ns), _ibr(nd), _ibr(sij), _ibr(dij), _p(vls), _p(ss), _p(sds), _p(posn), _p(vld), _p(sd), _p(sdd),
            _ibr(1 if xdown else 0), _ibr(1 if xup else 0), ctypes.byref(R(gdown)), ctypes.byref(R(gup))]
    if extended:
        dummy = np.zeros((1, 1), vls.dtype)
        args += [None, _ibr(0), _ibr(0), _p(dummy), _p(dummy)]
    f(*args)


def interp(algo, n, vls, ss, sds, posn, vld, sd, sdd, xdown=False, xup=False, extended=False):
    fortran_apply(INTERP_SYMBOL[algo], n, vls, ss, sds, posn, vld, sd, sdd, xdown, xup, 0.0, 0.0, extended)


def extrap(kind, n, vls, ss, sds, posn, vld, sd, sdd, xdown, xup, gdown=0.0, gup=0.0, extended=False):
    fortran_apply(EXTRAP_SYMBOL[kind], n, vls, ss, sds, posn, vld, sd, sdd, xdown, xup, gdown, gup, extended)


# ---------------------------------------------------------------- device tensors
def _dims(vls, vld):
    ns, sij = vls.shape; nd, dij = vld.shape
    return ns, sij, nd, dij


def findpos_dev(n, vls, posn, vld):
    ns, sij, nd, dij = _dims(vls, vld)
    return _lib().interpv_hip_findpos_dev(_prec8(vls), n, ns, nd, sij, dij, _p(vls), _p(posn), _p(vld))


def interp_dev(algo, n, vls, ss, sds, posn, vld, sd, sdd, xdown=False, xup=False):
    ns, sij, nd, dij = _dims(vls, vld)
    return _lib().interpv_hip_interp_dev(_prec8(vls), algo, n, ns, nd, sij, dij, _p(vls), _p(ss), _p(sds), _p(posn), _p(vld), _p(sd), _p(sdd),
                                         int(xdown), int(xup))


def extrap_dev(kind, n, vls, ss, sds, posn, vld, sd, sdd, xdown, xup, gdown=0.0, gup=0.0):
    ns, sij, nd, dij = _dims(vls, vld)
    return _lib().interpv_hip_extrap_dev(_prec8(vls), kind, n, ns, nd, sij, dij, _p(vls), _p(ss), _p(sds), _p(posn), _p(vld), _p(sd), _p(sdd),
                                         int(xdown), int(xup), float(gdown), float(gup))


def column_dev(algo, kind, n, vls, ss, sds, posn, vld, sd, sdd, xdown=False, xup=False, gdown=0.0, gup=0.0):
    """FindPos -> algo -> extrapolation in one pass; posn may be None (brackets not stored)"""
    ns, sij, nd, dij = _dims(vls, vld)
    return _lib().interpv_hip_column_dev(_prec8(vls), algo, kind, n, ns, nd, sij, dij, _p(vls), _p(ss), _p(sds), _p(posn), _p(vld), _p(sd), _p(sdd),
                                         int(xdown), int(xup), float(gdown), float(gup))
